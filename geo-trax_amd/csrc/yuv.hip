// Planar YUV 4:2:0 (I420) -> packed BGR u8: the colour conversion a decoder applies before the reference's loop gets
// its frame (cv2.VideoCapture.read() hands geotrax/extract.py:146 a BGR ndarray).
// Arithmetic: ITU-R BT.601 limited range in the 20-bit fixed point OpenCV's cvtColor(COLOR_YUV2BGR_I420) publishes
// (color_yuv.simd.hpp: CY 1220542, CUB 2116026, CUG -409993, CVG -852492, CVR 1673527, shift 20), one chroma sample per
// 2x2 luma block (no chroma interpolation). HBM-bound: 1.5 B read + 3 B written per pixel; a thread converts a 4x2
// block (8 luma bytes, 2+2 chroma bytes in, two 12-byte runs out), so a wave writes 768-byte runs per row.
#include <hip/hip_runtime.h>

#include "detector.hpp"
#include "geometry.hpp"

namespace gtx {
namespace {
constexpr int kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527, kShift = 20;

__device__ __forceinline__ uint32_t sat8(int v) { return (uint32_t)min(max(v, 0), 255); }

__global__ __launch_bounds__(256) void yuv420_to_bgr_kernel(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ up,
                                                            const uint8_t* __restrict__ vp, uint8_t* __restrict__ bgr, int h, int w) {
  const int bx = (blockIdx.x * blockDim.x + threadIdx.x) * 4, by = blockIdx.y * 2;     // top-left luma of this thread's 4x2 block
  if (bx >= w || by >= h) return;
  const int cw = (w + 1) >> 1;
  const bool full = bx + 4 <= w && ((w & 3) == 0);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int y = by + r;
    if (y >= h) break;
    uint32_t out[3] = {0, 0, 0};
    uint8_t yy[4];
    if (full) {
      const uint32_t q = *reinterpret_cast<const uint32_t*>(yp + (size_t)y * w + bx);
      yy[0] = q & 255; yy[1] = (q >> 8) & 255; yy[2] = (q >> 16) & 255; yy[3] = q >> 24;
    } else {
      for (int k = 0; k < 4; ++k) yy[k] = bx + k < w ? yp[(size_t)y * w + bx + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cx = min((bx + k) >> 1, cw - 1);
      const int u = (int)up[(size_t)(by >> 1) * cw + cx] - 128, v = (int)vp[(size_t)(by >> 1) * cw + cx] - 128;
      const int yl = max(0, (int)yy[k] - 16) * kCY;
      const int ruv = (1 << (kShift - 1)) + kCVR * v, guv = (1 << (kShift - 1)) + kCVG * v + kCUG * u, buv = (1 << (kShift - 1)) + kCUB * u;
      const uint32_t b = sat8((yl + buv) >> kShift), g = sat8((yl + guv) >> kShift), rr = sat8((yl + ruv) >> kShift);
      const int o = 3 * k;
      out[o >> 2] |= b << (8 * (o & 3));
      out[(o + 1) >> 2] |= g << (8 * ((o + 1) & 3));
      out[(o + 2) >> 2] |= rr << (8 * ((o + 2) & 3));
    }
    uint8_t* d = bgr + ((size_t)y * w + bx) * 3;
    if (full) {
      uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
      d4[0] = out[0]; d4[1] = out[1]; d4[2] = out[2];
    } else {
      for (int k = 0; k < 12 && bx + k / 3 < w; ++k) d[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
    }
  }
}
}  // namespace

// yuv: one I420 frame in HBM (Y plane h*w, then U and V planes ((h+1)/2)*((w+1)/2) each); bgr: h*w*3. Asynchronous.
void yuv420_to_bgr_dev(gtx_ctx* ctx, const void* yuv, int h, int w, void* bgr) {
  GTX_HIP(hipSetDevice(ctx->device));
  GTX_CHECK(h > 0 && w > 0, "yuv420_to_bgr: bad size %dx%d", w, h);
  const uint8_t* y = static_cast<const uint8_t*>(yuv);
  const size_t csz = (size_t)((h + 1) / 2) * ((w + 1) / 2);
  const uint8_t* u = y + (size_t)h * w;
  hipLaunchKernelGGL(yuv420_to_bgr_kernel, dim3(cdiv(cdiv(w, 4), 256), cdiv(h, 2)), dim3(256), 0, ctx->stream, y, u, u + csz,
                     static_cast<uint8_t*>(bgr), h, w);
  GTX_HIP(hipGetLastError());
}
}  // namespace gtx
