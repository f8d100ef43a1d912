// CLAHE for the stabilizer (stabilo `clahe: true`, reference geotrax/cfg/default.yaml:105; the `stable` preset).
// Compiled with -ffp-contract=off (Makefile): the float32 blend below is OpenCV's operation sequence, and a fused
// multiply-add changes which way the exact .5 cases round (hipcc fuses through __fmul_rn / __fadd_rn and through a
// function-level `#pragma clang fp contract(off)`; 30 of 2 M pixels differed by one grey level at 1080p).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "stabilizer.hpp"

namespace gtx {

namespace {
// cv2.createCLAHE(clipLimit = 2.0, tileGridSize = (8, 8)).apply() on the working-resolution gray image, as OpenCV
// publishes it (imgproc/src/clahe.cpp): per-tile clipped histogram -> LUT, then the four surrounding LUTs blended
// bilinearly per pixel in float32 (no fused multiply-adds: the operation sequence is OpenCV's), round half to even.
constexpr int kClaheGrid = 8;
__device__ __forceinline__ int refl101_near(int i, int n) { return i >= n ? 2 * n - 2 - i : i; }

// One workgroup per tile. The image is extended (reflect-101) to a multiple of the grid for the histograms only.
__global__ __launch_bounds__(256) void clahe_lut_kernel(const uint8_t* __restrict__ src, int w, int h, int tw, int th, int clip,
                                                        float lut_scale, uint8_t* __restrict__ luts) {
  __shared__ int s_hist[256];
  __shared__ int s_clipped;
  const int tx = blockIdx.x % kClaheGrid, ty = blockIdx.x / kClaheGrid, tid = threadIdx.x;
  s_hist[tid] = 0;
  if (tid == 0) s_clipped = 0;
  __syncthreads();
  for (int i = tid; i < tw * th; i += 256) {
    const int y = refl101_near(ty * th + i / tw, h), x = refl101_near(tx * tw + i % tw, w);
    atomicAdd(&s_hist[src[(size_t)y * w + x]], 1);
  }
  __syncthreads();
  int v = s_hist[tid];
  if (clip > 0) {
    if (v > clip) { atomicAdd(&s_clipped, v - clip); v = clip; }
    __syncthreads();
    const int clipped = s_clipped, batch = clipped / 256, residual = clipped - batch * 256;
    v += batch;
    if (residual != 0) {
      const int step = max(256 / residual, 1);        // one extra count on every step-th bin from bin 0, `residual` times
      if (tid % step == 0 && tid / step < residual) v += 1;
    }
  }
  s_hist[tid] = v;
  __syncthreads();
  if (tid == 0) {
    int sum = 0;
    for (int i = 0; i < 256; ++i) { sum += s_hist[i]; s_hist[i] = sum; }
  }
  __syncthreads();
  const int r = __float2int_rn(__fmul_rn((float)s_hist[tid], lut_scale));
  luts[blockIdx.x * 256 + tid] = (uint8_t)min(max(r, 0), 255);
}

__global__ __launch_bounds__(256) void clahe_apply_kernel(const uint8_t* __restrict__ src, int w, int h, float inv_tw, float inv_th,
                                                          const uint8_t* __restrict__ luts, uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const float txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f);      // inv_tw = 1.0f / tile width, divided on the host
  const float tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f);
  int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
  const float xa = __fsub_rn(txf, (float)tx1), ya = __fsub_rn(tyf, (float)ty1);
  const float xa1 = __fsub_rn(1.0f, xa), ya1 = __fsub_rn(1.0f, ya);
  const int tx2 = min(tx1 + 1, kClaheGrid - 1), ty2 = min(ty1 + 1, kClaheGrid - 1);
  tx1 = max(tx1, 0); ty1 = max(ty1, 0);
  const int v = src[(size_t)y * w + x];
  const float l11 = luts[(ty1 * kClaheGrid + tx1) * 256 + v], l12 = luts[(ty1 * kClaheGrid + tx2) * 256 + v];
  const float l21 = luts[(ty2 * kClaheGrid + tx1) * 256 + v], l22 = luts[(ty2 * kClaheGrid + tx2) * 256 + v];
  const float top = __fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), bot = __fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa));
  const int r = __float2int_rn(__fadd_rn(__fmul_rn(top, ya1), __fmul_rn(bot, ya)));
  dst[(size_t)y * w + x] = (uint8_t)min(max(r, 0), 255);
}

}  // namespace

// src, dst: u8 [h][w] in HBM (may be the same buffer); luts: kClaheLutBytes of scratch.
void clahe_dev(const uint8_t* src, int h, int w, uint8_t* luts, uint8_t* dst, hipStream_t s) {
  // OpenCV extends BOTH axes by grid - (size % grid) unless both divide: an axis that does divide then grows by a whole grid step.
  const bool exact = w % kClaheGrid == 0 && h % kClaheGrid == 0;
  const int ew = exact ? w : w + (kClaheGrid - w % kClaheGrid), eh = exact ? h : h + (kClaheGrid - h % kClaheGrid);
  const int tw = ew / kClaheGrid, th = eh / kClaheGrid, area = tw * th;
  hipLaunchKernelGGL(clahe_lut_kernel, dim3(kClaheGrid * kClaheGrid), dim3(256), 0, s, src, w, h, tw, th, std::max((int)(2.0 * area / 256), 1),
                     255.0f / (float)area, luts);
  hipLaunchKernelGGL(clahe_apply_kernel, dim3((w + 255) / 256, h), dim3(256), 0, s, src, w, h, 1.0f / (float)tw, 1.0f / (float)th, luts, dst);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
