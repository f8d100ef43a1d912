// Perspective frame warp: what cv2.warpPerspective(frame, H, (w, h)) computes for the reference's
// visualisation modes 1 and 4 (geotrax/visualize.py:285-289):
//     dst(x, y) = bilinear src(H^-1 (x, y)),  INTER_LINEAR, BORDER_CONSTANT 0.
// Arithmetic follows OpenCV's published scheme (imgwarp.cpp, warpPerspective + remap with INTER_BITS = 5):
// the source coordinate is formed in float64 as (X0 + M0*x1) * (32 / W) with X0 = M0*bx + M1*y + M2 per
// 64-pixel block, rounded to nearest (ties to even) into 1/32-pixel units, and the four neighbours are blended
// with the integer weights (32-ax)(32-ay) ... ax*ay (x32 = OpenCV's 15-bit table entries), + half, >> 10.
// This file is compiled with -ffp-contract=off so that the float64 operation sequence is the stated one.
//
// HBM-bound by construction (24.9 MB in + 24.9 MB out per 4K frame). One workgroup = 128 x 8 destination
// pixels; a thread owns 4 consecutive pixels and stores them as 12 contiguous bytes (a wave writes two 384-B
// runs). The source footprint of the tile (its bounding box, a few KB for any near-identity camera motion) is
// copied into LDS with 16-byte loads first; tiles whose footprint does not fit (strong zoom-out / rotation) take
// the direct path, which gathers from global memory.
#include <hip/hip_runtime.h>

#include "detector.hpp"
#include "geometry.hpp"

namespace gtx {

namespace {
struct Mat3 { double m[9]; };

constexpr int kTW = 128, kTH = 8;        // destination tile
constexpr int kLdsCap = 16 * 1024;       // bytes of source footprint staged per workgroup (a 128x8 tile under a near-identity
                                         // homography needs ~4-5 KB; 16 KB keeps 8 workgroups per CU)

// Row terms of OpenCV's coordinate arithmetic: X0 = M0*bx + M1*y + M2 at the start of the 64-pixel block that holds x.
struct RowTerms { double X0, Y0, W0; };
__device__ __forceinline__ RowTerms row_terms(const Mat3& M, int bx, int y) {
  RowTerms r;
  r.X0 = M.m[0] * bx + M.m[1] * y + M.m[2];
  r.Y0 = M.m[3] * bx + M.m[4] * y + M.m[5];
  r.W0 = M.m[6] * bx + M.m[7] * y + M.m[8];
  return r;
}
// Source coordinate of pixel bx + x1 in 1/32-pixel units: saturate_cast<int>((X0 + M0*x1) * (32 / W)).
__device__ __forceinline__ void src_coord(const Mat3& M, const RowTerms& r, int x1, int& X, int& Y) {
  double W = r.W0 + M.m[6] * x1;
  W = W != 0.0 ? 32.0 / W : 0.0;
  const double fX = fmax(-2147483648.0, fmin(2147483647.0, (r.X0 + M.m[0] * x1) * W));
  const double fY = fmax(-2147483648.0, fmin(2147483647.0, (r.Y0 + M.m[3] * x1) * W));
  X = (int)rint(fX);                     // |fX| <= 2^31 after the clamp; rint = round half to even like cvRound
  Y = (int)rint(fY);
}

__global__ __launch_bounds__(256) void warp_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int h, int w, const Mat3 M) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsCap];
  const int tid = threadIdx.x, lane = tid & 63;
  const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * kTH;
  // Footprint of the tile: source coordinates of its four corners (a projective map with W > 0 over the tile sends the
  // tile to a convex quad whose extremes are at the corners). Every wave computes it for itself -- lanes 0..3 take one
  // corner each, a butterfly over those four lanes combines them -- so no barrier is needed before the staging loop.
  int bx0, by0, bx1, by1;
  bool staged;
  {
    const int ci = lane & 1, cj = (lane >> 1) & 1;
    const int cx = ci ? min(tx0 + kTW, w) - 1 : tx0, cy = cj ? min(ty0 + kTH, h) - 1 : ty0;
    int X, Y;
    src_coord(M, row_terms(M, cx & ~63, cy), cx & 63, X, Y);
    int lo_x = X >> 5, hi_x = lo_x, lo_y = Y >> 5, hi_y = lo_y;
    int pos = (M.m[6] * cx + M.m[7] * cy + M.m[8]) > 1e-9 ? 1 : 0;
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) {
      lo_x = min(lo_x, __shfl_xor(lo_x, d)); hi_x = max(hi_x, __shfl_xor(hi_x, d));
      lo_y = min(lo_y, __shfl_xor(lo_y, d)); hi_y = max(hi_y, __shfl_xor(hi_y, d));
      pos &= __shfl_xor(pos, d);
    }
    lo_x = __shfl(lo_x, 0); hi_x = __shfl(hi_x, 0); lo_y = __shfl(lo_y, 0); hi_y = __shfl(hi_y, 0); pos = __shfl(pos, 0);
    // one pixel of margin on each side for the second bilinear tap and the per-pixel rounding
    bx0 = max(0, min(w, lo_x - 1)); bx1 = max(0, min(w, hi_x + 3));
    by0 = max(0, min(h, lo_y - 1)); by1 = max(0, min(h, hi_y + 3));
    if (lo_x > w || hi_x < -4) { bx0 = bx1 = 0; }
    if (lo_y > h || hi_y < -4) { by0 = by1 = 0; }
    const long row_bytes = bx1 > bx0 ? (((long)bx1 * 3 + 15) & ~15L) - (((long)bx0 * 3) & ~15L) : 0;
    staged = pos && row_bytes > 0 && by1 > by0 && row_bytes * (by1 - by0) <= kLdsCap;
  }
  const int a0 = (bx0 * 3) & ~15;                                 // byte offset in a source row where the staged run starts
  const int pitch = staged ? ((bx1 * 3 + 15) & ~15) - a0 : 0;
  if (staged) {                                                   // (uniform over the workgroup: every wave derived the same box)
    const int chunks = pitch >> 4, total = chunks * (by1 - by0);
    const size_t row_len = (size_t)w * 3, img_len = (size_t)h * row_len;
    const bool aligned = (row_len & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    for (int i = tid; i < total; i += 256) {
      const int r = i / chunks, c = i - r * chunks;
      const size_t off = (size_t)(by0 + r) * row_len + a0 + (size_t)c * 16;
      uint4 v;
      if (aligned && off + 16 <= img_len) {
        v = *reinterpret_cast<const uint4*>(src + off);
      } else {                                                    // image tail, or rows that are not 16-byte multiples
        uint32_t q[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; ++k)
          if (off + k < img_len) q[k >> 2] |= (uint32_t)src[off + k] << (8 * (k & 3));
        v = make_uint4(q[0], q[1], q[2], q[3]);
      }
      *reinterpret_cast<uint4*>(lds + r * pitch + c * 16) = v;
    }
    __syncthreads();
  }
  const int y = ty0 + (tid >> 5), xq = tx0 + 4 * (tid & 31);
  if (y >= h || xq >= w) return;
  const RowTerms rt = row_terms(M, xq & ~63, y);                  // the thread's four pixels share a 64-pixel block and a row
  uint32_t out[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int X, Y;
    src_coord(M, rt, (xq & 63) + k, X, Y);
    const int x0 = X >> 5, y0 = Y >> 5;
    const int ax = X & 31, ay = Y & 31;
    // separable form of the four-tap blend (exact in integers): horizontal with (32-ax, ax), vertical with (32-ay, ay)
    int top[3], bot[3];
    if (staged && x0 >= bx0 && x0 + 1 < bx1 && y0 >= by0 && y0 + 1 < by1) {   // all four taps inside the staged box (and the image)
      const uint8_t* p = lds + (y0 - by0) * pitch + (x0 * 3 - a0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        top[c] = p[c] * (32 - ax) + p[3 + c] * ax;
        bot[c] = p[pitch + c] * (32 - ax) + p[pitch + 3 + c] * ax;
      }
    } else {
      auto at = [&](int yy, int xx, int c) -> int {
        return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? src[((size_t)yy * w + xx) * 3 + c] : 0;
      };
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        top[c] = at(y0, x0, c) * (32 - ax) + at(y0, x0 + 1, c) * ax;
        bot[c] = at(y0 + 1, x0, c) * (32 - ax) + at(y0 + 1, x0 + 1, c) * ax;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const uint32_t v = (uint32_t)((top[c] * (32 - ay) + bot[c] * ay + 512) >> 10);
      const int byte = 3 * k + c;
      out[byte >> 2] |= v << (8 * (byte & 3));
    }
  }
  uint8_t* d = dst + ((size_t)y * w + xq) * 3;
  if (xq + 4 <= w && ((reinterpret_cast<uintptr_t>(d) & 3) == 0)) {
    uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
    d4[0] = out[0]; d4[1] = out[1]; d4[2] = out[2];
  } else {
    for (int k = 0; k < 12 && xq + k / 3 < w; ++k) d[k] = (uint8_t)(out[k >> 2] >> (8 * (k & 3)));
  }
}
}  // namespace

void warp_frame_dev(gtx_ctx* ctx, const void* src_bgr, int h, int w, const double H[9], void* dst_bgr) {
  GTX_HIP(hipSetDevice(ctx->device));
  GTX_CHECK(h > 0 && w > 0, "warp_frame: bad size %dx%d", w, h);
  Mat3 Hi;
  if (!invert3x3(H, Hi.m)) fail(-1, "warp_frame: homography is singular");
  hipLaunchKernelGGL(warp_kernel, dim3(cdiv(w, kTW), cdiv(h, kTH)), dim3(256), 0, ctx->stream, static_cast<const uint8_t*>(src_bgr),
                     static_cast<uint8_t*>(dst_bgr), h, w, Hi);
  GTX_HIP(hipGetLastError());
}

void warp_frame(gtx_ctx* ctx, const uint8_t* src_bgr, int h, int w, const double H[9], uint8_t* dst_bgr) {
  GTX_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)h * w * 3;
  DevBuf ds(bytes), dd(bytes);
  GTX_HIP(hipMemcpyAsync(ds.p, src_bgr, bytes, hipMemcpyHostToDevice, ctx->stream));
  warp_frame_dev(ctx, ds.p, h, w, H, dd.p);
  GTX_HIP(hipMemcpyAsync(dst_bgr, dd.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GTX_HIP(hipStreamSynchronize(ctx->stream));
}

}  // namespace gtx
