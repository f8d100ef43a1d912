// Perspective frame warp (cv2.warpPerspective equivalent used by the reference's visualisation
// modes, geotrax/visualize.py:289): dst(x,y) = bilinear src(H^-1 (x,y)), constant 0 outside.
// HBM-bound: one thread per destination pixel, 3 channels, reads hit L2 (the source footprint of
// a wave is a thin strip).
#include <hip/hip_runtime.h>

#include "detector.hpp"
#include "geometry.hpp"

namespace gtx {

namespace {
struct Mat3 { double m[9]; };

__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int h, int w, const Mat3 Hi) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const double den = Hi.m[6] * x + Hi.m[7] * y + Hi.m[8];
  uint8_t out[3] = {0, 0, 0};
  if (fabs(den) > 1e-12) {
    const double sx = (Hi.m[0] * x + Hi.m[1] * y + Hi.m[2]) / den, sy = (Hi.m[3] * x + Hi.m[4] * y + Hi.m[5]) / den;
    // OpenCV quantises the source coordinate to 1/32 px (INTER_BITS = 5) and blends with
    // 15-bit fixed-point weights; the same quantisation is used here.
    const long fx = llrint(sx * 32.0), fy = llrint(sy * 32.0);
    const int x0 = (int)(fx >> 5), y0 = (int)(fy >> 5);
    const int ax = (int)(fx & 31), ay = (int)(fy & 31);
    const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      auto at = [&](int yy, int xx) -> int {
        return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? src[((size_t)yy * w + xx) * 3 + c] : 0;
      };
      const int v = at(y0, x0) * w00 + at(y0, x0 + 1) * w01 + at(y0 + 1, x0) * w10 + at(y0 + 1, x0 + 1) * w11;
      out[c] = (uint8_t)((v + 512) >> 10);
    }
  }
  uint8_t* d = dst + ((size_t)y * w + x) * 3;
  d[0] = out[0]; d[1] = out[1]; d[2] = out[2];
}
}  // namespace

void warp_frame(gtx_ctx* ctx, const uint8_t* src_bgr, int h, int w, const double H[9], uint8_t* dst_bgr) {
  GTX_HIP(hipSetDevice(ctx->device));
  Mat3 Hi;
  if (!invert3x3(H, Hi.m)) fail(-1, "warp_frame: homography is singular");
  const size_t bytes = (size_t)h * w * 3;
  DevBuf ds(bytes), dd(bytes);
  GTX_HIP(hipMemcpyAsync(ds.p, src_bgr, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(warp_kernel, dim3(cdiv(w, 256), h), dim3(256), 0, ctx->stream, ds.as<uint8_t>(), dd.as<uint8_t>(), h, w, Hi);
  GTX_HIP(hipGetLastError());
  GTX_HIP(hipMemcpyAsync(dst_bgr, dd.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GTX_HIP(hipStreamSynchronize(ctx->stream));
}

}  // namespace gtx
