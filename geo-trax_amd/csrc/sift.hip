// SIFT / RootSIFT on the GPU (gfx950). Replaces what cv2.SIFT_create(nfeatures,
// enable_precise_upscale=True).detectAndCompute + stabilo's RootSIFT conversion compute for the
// orthophoto / master-frame registration (reference: geotrax/utils/registration.py:59-85 with
// detector_name='rsift'). Algorithm: Lowe's SIFT with OpenCV's constants
// (sigma 1.6, 3 layers per octave, contrast 0.04, edge 10, image doubled first, 36-bin orientation
// histogram, 4x4x8 descriptor clamped at 0.2 and quantised to 0..255); oracle/sift_ref.py is the
// line-by-line CPU restatement the parity tests compare against.
//
// This file is compiled with -ffp-contract=off: the Gaussian pyramid, the DoG images and the
// sub-pixel refinement are plain float32 operation sequences in a fixed order, so they are
// bit-identical to the oracle's numpy float32 arithmetic. The two histogram stages accumulate
// in float64 (LDS atomics), which makes them independent of the summation order to float32
// precision; they use expf/atan2f, whose last-bit differences from libm are the only source of
// (rare, quantisation-level) descriptor differences.
//
// Memory: the whole Gaussian and DoG pyramids of the doubled image stay resident in HBM
// (11 float images per octave, ~59 B per doubled pixel: 1.95 GB for a 4K frame, ~35 GB for a
// 15000 x 10000 orthophoto cut-out) -- sized for 288 GB, nothing is recomputed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>

#include "sift.hpp"

namespace gtx {

namespace {

constexpr int kLayers = 3, kGauss = kLayers + 3, kDog = kLayers + 2;
constexpr int kBorder = 5, kMaxInterp = 5, kOriBins = 36, kMaxOct = 16;
constexpr float kContrastThr = 0.04f, kEdgeThr = 10.0f;
constexpr double kSigma = 1.6;

struct OctaveTable {
  float* g[kMaxOct][kGauss];
  float* d[kMaxOct][kDog];
  int w[kMaxOct], h[kMaxOct];
  int n;
};

struct Cand { int o, layer, r, c; };
struct Refined {
  float x, y, size, response;
  int word, o, layer, r, c;
  Cand key;
};
struct Oriented {
  float x, y, size, angle, response;
  int word, o, layer;
  Cand key;
  int bin;
};
struct Final {              // what the descriptor kernel needs
  double ori;               // 360 - angle, degrees
  float px, py;             // octave-local position
  float scl;
  int o, layer;
  int pad;
};

__device__ __forceinline__ int refl101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i >= n ? p - i : i;
}

__global__ void gray_kernel(const uint8_t* __restrict__ bgr, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
  out[i] = (float)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
}

// dst(x, y) = bilinear(src, x / 2, y / 2), replicate past the last row / column
__global__ void upscale_kernel(const float* __restrict__ g, int h, int w, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= 2 * w) return;
  const int sx = x >> 1, sy = y >> 1, sx1 = min(sx + 1, w - 1), sy1 = min(sy + 1, h - 1);
  const float a = g[(size_t)sy * w + sx], b = g[(size_t)sy * w + sx1], c = g[(size_t)sy1 * w + sx], d = g[(size_t)sy1 * w + sx1];
  float v;
  if ((x & 1) == 0 && (y & 1) == 0) v = a;
  else if ((y & 1) == 0) v = (a + b) * 0.5f;
  else if ((x & 1) == 0) v = (a + c) * 0.5f;
  else v = ((a + b) + (c + d)) * 0.25f;
  out[(size_t)y * 2 * w + x] = v;
}

constexpr int kMaxRadius = 16;
struct Taps { float w[kMaxRadius + 1]; int r; };   // w[0] centre, w[k] = tap at +-k

__global__ __launch_bounds__(256) void blur_h_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, const Taps t) {
  __shared__ float s[256 + 2 * kMaxRadius];
  const int y = blockIdx.y, x0 = blockIdx.x * 256;
  const float* row = src + (size_t)y * w;
  for (int i = threadIdx.x; i < 256 + 2 * t.r; i += 256) s[i] = row[refl101(x0 - t.r + i, w)];
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= w) return;
  const float* c = s + t.r + threadIdx.x;
  float acc = c[0] * t.w[0];
  for (int k = 1; k <= t.r; ++k) acc = acc + t.w[k] * (c[-k] + c[k]);
  dst[(size_t)y * w + x] = acc;
}

__global__ __launch_bounds__(256) void blur_v_kernel(const float* __restrict__ src, float* __restrict__ dst, int w, int h, const Taps t) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  float acc = src[(size_t)y * w + x] * t.w[0];
  for (int k = 1; k <= t.r; ++k)
    acc = acc + t.w[k] * (src[(size_t)refl101(y - k, h) * w + x] + src[(size_t)refl101(y + k, h) * w + x]);
  dst[(size_t)y * w + x] = acc;
}

// Both passes of a Gaussian blur for one 64 x 32 tile of the output, and the difference-of-Gaussians layer with it: the tile's source
// pixels with their halo (refl101 at the image border) go into LDS once, the horizontal pass fills a second LDS image for the tile's rows
// + halo rows, the vertical pass reads that and writes the blurred pixel and, when `dog` is given, blurred - source (the source of
// Gaussian layer i + 1 IS layer i: DoG layer i costs one more store instead of a pass of its own). Same products and sums in the same
// order as blur_h_kernel / blur_v_kernel / sub_kernel (the file is built with -ffp-contract=off): bit-identical images, 12-17 bytes of
// HBM traffic per pixel and layer instead of 28 plus a (2 r + 1)-fold re-read of the scratch image through the caches.
constexpr int kBlurTW = 64, kBlurTH = 32;
__global__ __launch_bounds__(256) void blur_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int w, int h,
                                                        const Taps t) {
  __shared__ float sA[kBlurTH + 2 * kMaxRadius][kBlurTW + 2 * kMaxRadius + 1];
  __shared__ float sB[kBlurTH + 2 * kMaxRadius][kBlurTW];
  const int r = t.r, x0 = blockIdx.x * kBlurTW, y0 = blockIdx.y * kBlurTH;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int ry = ty; ry < kBlurTH + 2 * r; ry += 4) {
    const float* row = src + (size_t)refl101(y0 - r + ry, h) * w;
    for (int rx = tx; rx < kBlurTW + 2 * r; rx += 64) sA[ry][rx] = row[refl101(x0 - r + rx, w)];
  }
  __syncthreads();
  for (int ry = ty; ry < kBlurTH + 2 * r; ry += 4) {
    const float* c = &sA[ry][tx + r];
    float acc = c[0] * t.w[0];
    for (int k = 1; k <= r; ++k) acc = acc + t.w[k] * (c[-k] + c[k]);
    sB[ry][tx] = acc;
  }
  __syncthreads();
  const int x = x0 + tx;
  for (int oy = ty; oy < kBlurTH; oy += 4) {
    const int y = y0 + oy;
    if (x >= w || y >= h) continue;
    float acc = sB[oy + r][tx] * t.w[0];
    for (int k = 1; k <= r; ++k) acc = acc + t.w[k] * (sB[oy + r - k][tx] + sB[oy + r + k][tx]);
    dst[(size_t)y * w + x] = acc;
    if (dog) dog[(size_t)y * w + x] = acc - sA[oy + r][tx + r];
  }
}

// The same tile with the radius known at compile time (the five radii of the default scale space: 5, 6, 8, 10, 13): both passes keep a
// sliding window in registers -- a thread makes 4 neighbouring outputs of a row from 4 + 2 R staged values, then 8 outputs down a column
// from 8 + 2 R -- instead of 2 R + 1 LDS reads per output, and the tap loops unroll (the generic kernel above issued its LDS reads one
// dependent iteration at a time: 21 ps per pixel on the 30 000 x 30 000 base octave). Same sums in the same order.
template <int R>
__global__ __launch_bounds__(256) void blur_tile_kernel_r(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int w, int h,
                                                          const Taps t) {
  constexpr int TW = kBlurTW, TH = kBlurTH, AW = TW + 2 * R, AH = TH + 2 * R;
  __shared__ float sA[AH][AW + 1];
  __shared__ float sB[AH][TW + 1];
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  float wk[R + 1];
#pragma unroll
  for (int k = 0; k <= R; ++k) wk[k] = t.w[k];
  for (int i = threadIdx.x; i < AH * AW; i += 256) {
    const int ry = i / AW, rx = i - ry * AW;              // AW is a compile-time constant: a multiply-shift, no division
    sA[ry][rx] = src[(size_t)refl101(y0 - R + ry, h) * w + refl101(x0 - R + rx, w)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < AH * (TW / 4); i += 256) {   // horizontal pass: 4 outputs per item
    const int g4 = i / AH, ry = i - g4 * AH, gx = g4 * 4;    // neighbouring lanes take neighbouring ROWS: the pitch (AW + 1) is odd for every R
    float win[4 + 2 * R];                                   // here, so a window element is read from 64 different banks
#pragma unroll
    for (int k = 0; k < 4 + 2 * R; ++k) win[k] = sA[ry][gx + k];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float acc = win[o + R] * wk[0];
#pragma unroll
      for (int k = 1; k <= R; ++k) acc = acc + wk[k] * (win[o + R - k] + win[o + R + k]);
      sB[ry][gx + o] = acc;
    }
  }
  __syncthreads();
  {                                                          // vertical pass: 8 outputs down a column per thread
    const int tx = threadIdx.x & 63, oy0 = (threadIdx.x >> 6) * 8;
    float win[8 + 2 * R];
#pragma unroll
    for (int k = 0; k < 8 + 2 * R; ++k) win[k] = sB[oy0 + k][tx];
    const int x = x0 + tx;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float acc = win[o + R] * wk[0];
#pragma unroll
      for (int k = 1; k <= R; ++k) acc = acc + wk[k] * (win[o + R - k] + win[o + R + k]);
      const int y = y0 + oy0 + o;
      if (x < w && y < h) {
        dst[(size_t)y * w + x] = acc;
        if (dog) dog[(size_t)y * w + x] = acc - sA[oy0 + o + R][tx + R];
      }
    }
  }
}

__global__ void down_kernel(const float* __restrict__ src, int sh, int sw, float* __restrict__ dst, int dh, int dw, double fy, double fx) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  const int sx = min((int)floor(x * fx), sw - 1), sy = min((int)floor(y * fy), sh - 1);
  dst[(size_t)y * dw + x] = src[(size_t)sy * sw + sx];
}

__global__ void sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] - b[i];
}

__global__ __launch_bounds__(256) void extrema_kernel(const float* __restrict__ prv, const float* __restrict__ cur, const float* __restrict__ nxt,
                                                      int w, int h, float thr, int o, int layer, Cand* __restrict__ cand,
                                                      int* __restrict__ n_cand, int cap) {
  const int x = kBorder + blockIdx.x * 256 + threadIdx.x, y = kBorder + blockIdx.y;
  bool is_ext = x < w - kBorder && y < h - kBorder;
  if (is_ext) {
    const size_t p = (size_t)y * w + x;
    const float v = cur[p];
    is_ext = fabsf(v) > thr;
    if (is_ext) {
      float mx = -INFINITY, mn = INFINITY;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const size_t q = p + (long)dy * w + dx;
          const float a = prv[q], b = nxt[q];
          mx = fmaxf(mx, fmaxf(a, b)); mn = fminf(mn, fminf(a, b));
          if (dy != 0 || dx != 0) { const float c = cur[q]; mx = fmaxf(mx, c); mn = fminf(mn, c); }
        }
      is_ext = (v > 0.f && v >= mx) || (v < 0.f && v <= mn);
    }
  }
  const int slot = gtx_wave_append(n_cand, is_ext);
  if (is_ext && slot < cap) cand[slot] = Cand{o, layer, y, x};
}

// 3x3 solve, LU with partial pivoting, float32 (operation order of oracle/sift_ref.py::_solve3)
__device__ bool solve3(float A[3][3], float x[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int p = k;
    float best = fabsf(A[k][k]);
    for (int i = k + 1; i < 3; ++i)
      if (fabsf(A[i][k]) > best) { best = fabsf(A[i][k]); p = i; }
    if (best < 1.1920929e-7f) return false;
    if (p != k) {
      for (int j = 0; j < 3; ++j) { const float t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
      const float t = x[k]; x[k] = x[p]; x[p] = t;
    }
    for (int i = k + 1; i < 3; ++i) {
      const float f = A[i][k] / A[k][k];
      for (int j = k; j < 3; ++j) A[i][j] = A[i][j] - f * A[k][j];
      x[i] = x[i] - f * x[k];
    }
  }
  for (int k = 2; k >= 0; --k) {
    float s = x[k];
    for (int j = k + 1; j < 3; ++j) s = s - A[k][j] * x[j];
    x[k] = s / A[k][k];
  }
  return true;
}

// adjustLocalExtrema: thread per candidate
__global__ __launch_bounds__(256) void refine_kernel(const OctaveTable T, const Cand* __restrict__ cand, int n_cand, Refined* __restrict__ out,
                                                     int* __restrict__ n_out) {
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= n_cand) return;
  const Cand cd = cand[i0];
  const int o = cd.o, w = T.w[o], h = T.h[o];
  int layer = cd.layer, r = cd.r, c = cd.c;
  const float img_scale = 1.0f / 255.0f;
  const float ds = img_scale * 0.5f, sds = img_scale, cds = img_scale * 0.25f;
  float xi = 0.f, xr = 0.f, xc = 0.f;
  int it = 0;
#define AT(IMG, R, C) (IMG)[(size_t)(R) * w + (C)]
  for (; it < kMaxInterp; ++it) {
    const float* img = T.d[o][layer];
    const float* prv = T.d[o][layer - 1];
    const float* nxt = T.d[o][layer + 1];
    float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * ds, (AT(img, r + 1, c) - AT(img, r - 1, c)) * ds,
                   (AT(nxt, r, c) - AT(prv, r, c)) * ds};
    const float v2 = AT(img, r, c) * 2.f;
    const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * sds;
    const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * sds;
    const float dss = (AT(nxt, r, c) + AT(prv, r, c) - v2) * sds;
    const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cds;
    const float dxs = (AT(nxt, r, c + 1) - AT(nxt, r, c - 1) - AT(prv, r, c + 1) + AT(prv, r, c - 1)) * cds;
    const float dys = (AT(nxt, r + 1, c) - AT(nxt, r - 1, c) - AT(prv, r + 1, c) + AT(prv, r - 1, c)) * cds;
    float A[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
    if (!solve3(A, dD)) return;
    xi = -dD[2]; xr = -dD[1]; xc = -dD[0];
    if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
    if (fabsf(xi) > 1073741824.f || fabsf(xr) > 1073741824.f || fabsf(xc) > 1073741824.f) return;
    c += (int)rintf(xc); r += (int)rintf(xr); layer += (int)rintf(xi);
    if (layer < 1 || layer > kLayers || c < kBorder || c >= w - kBorder || r < kBorder || r >= h - kBorder) return;
  }
  if (it >= kMaxInterp) return;
  const float* img = T.d[o][layer];
  const float* prv = T.d[o][layer - 1];
  const float* nxt = T.d[o][layer + 1];
  const float d0 = (AT(img, r, c + 1) - AT(img, r, c - 1)) * ds, d1 = (AT(img, r + 1, c) - AT(img, r - 1, c)) * ds,
              d2 = (AT(nxt, r, c) - AT(prv, r, c)) * ds;
  const float t = (d0 * xc + d1 * xr) + d2 * xi;
  const float contr = AT(img, r, c) * img_scale + t * 0.5f;
  if (fabsf(contr) * (float)kLayers < kContrastThr) return;
  const float v2 = AT(img, r, c) * 2.f;
  const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * sds;
  const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * sds;
  const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cds;
#undef AT
  const float tr = dxx + dyy;
  const float det = dxx * dyy - dxy * dxy;
  if (det <= 0.f || (tr * tr) * kEdgeThr >= ((kEdgeThr + 1.f) * (kEdgeThr + 1.f)) * det) return;
  Refined R;
  const float scale = (float)(1 << o);
  R.x = ((float)c + xc) * scale;
  R.y = ((float)r + xr) * scale;
  R.word = o + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255.0) << 16);
  R.size = (float)(kSigma * exp2(((double)layer + (double)xi) / (double)kLayers) * (double)(1 << o) * 2.0);
  R.response = fabsf(contr);
  R.o = o; R.layer = layer; R.r = r; R.c = c;
  R.key = cd;
  const int slot = atomicAdd(n_out, 1);
  out[slot] = R;
}

// calcOrientationHist + peak picking: one wave per refined keypoint
__global__ __launch_bounds__(256) void orient_kernel(const OctaveTable T, const Refined* __restrict__ in, int n, Oriented* __restrict__ out,
                                                     int* __restrict__ n_out, int cap) {
  __shared__ double s_hist[4][kOriBins];
  __shared__ float s_tmp[4][kOriBins + 4], s_h[4][kOriBins];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;                       // whole wave exits together (i is wave-uniform)
  const Refined R = in[i];
  const int w = T.w[R.o], h = T.h[R.o];
  const float* img = T.g[R.o][R.layer];
  const double scl = (double)R.size * 0.5 / (double)(1 << R.o);
  const int radius = (int)rint(4.5 * scl);
  const double sigma = 1.5 * scl;
  const float expf_scale = (float)(-1.0 / (2.0 * sigma * sigma));
  if (lane < kOriBins) s_hist[wave][lane] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const int side = 2 * radius + 1;
  for (int k = lane; k < side * side; k += 64) {
    const int di = k / side - radius, dj = k % side - radius;
    const int y = R.r + di, x = R.c + dj;
    if (y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1) continue;
    const float dx = img[(size_t)y * w + x + 1] - img[(size_t)y * w + x - 1];
    const float dy = img[(size_t)(y - 1) * w + x] - img[(size_t)(y + 1) * w + x];
    const float wgt = expf((float)(di * di + dj * dj) * expf_scale);
    float ori = atan2f(dy, dx) * 57.29577951308232f;
    if (ori < 0.f) ori = ori + 360.f;
    const float mag = sqrtf(dx * dx + dy * dy);
    int b = (int)rintf(ori * (float)(kOriBins / 360.0));
    b = b % kOriBins;
    if (b < 0) b += kOriBins;
    atomicAdd(&s_hist[wave][b], (double)(wgt * mag));
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  if (lane < kOriBins) {
    const float v = (float)s_hist[wave][lane];
    s_tmp[wave][lane + 2] = v;
    if (lane < 2) s_tmp[wave][kOriBins + 2 + lane] = v;            // wrap: tmp[n], tmp[n+1]
    if (lane >= kOriBins - 2) s_tmp[wave][lane - (kOriBins - 2)] = v;  // tmp[-2], tmp[-1]
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float hv = -1.f;
  if (lane < kOriBins) {
    const float* t = s_tmp[wave] + lane;   // t[0..4] = tmp[lane-2 .. lane+2]
    hv = ((t[0] + t[4]) * (1.f / 16.f) + (t[1] + t[3]) * (4.f / 16.f)) + t[2] * (6.f / 16.f);
    s_h[wave][lane] = hv;
  }
  float mx = hv;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  if (lane < kOriBins) {
    const int j = lane, l = j > 0 ? j - 1 : kOriBins - 1, r2 = j < kOriBins - 1 ? j + 1 : 0;
    const float hl = s_h[wave][l], hr = s_h[wave][r2], hj = hv;
    if (hj > hl && hj > hr && hj >= mx * 0.8f) {
      double b = (double)j + 0.5 * (double)(hl - hr) / (double)(hl - 2.f * hj + hr);
      b = b < 0 ? kOriBins + b : (b >= kOriBins ? b - kOriBins : b);
      float a = (float)(360.0 - (360.0 / kOriBins) * b);
      if (fabs((double)a - 360.0) < 1.19e-7) a = 0.f;
      const int slot = atomicAdd(n_out, 1);
      if (slot < cap) {
        Oriented O;
        O.x = R.x; O.y = R.y; O.size = R.size; O.angle = a; O.response = R.response;
        O.word = R.word; O.o = R.o; O.layer = R.layer; O.key = R.key; O.bin = j;
        out[slot] = O;
      }
    }
  }
}

// calcSIFTDescriptor (+ RootSIFT): one wave per keypoint
constexpr int kD = 4, kN = 8, kHist = (kD + 2) * (kD + 2) * (kN + 2);
__global__ __launch_bounds__(256) void describe_kernel(const OctaveTable T, const Final* __restrict__ kps, int n, float* __restrict__ desc, int root,
                                                       float root_eps) {
  __shared__ double s_hist[4][kHist];
  __shared__ float s_dst[4][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n) return;
  const Final K = kps[i];
  const int w = T.w[K.o], h = T.h[K.o];
  const float* img = T.g[K.o][K.layer];
  for (int k = lane; k < kHist; k += 64) s_hist[wave][k] = 0.0;
  __builtin_amdgcn_wave_barrier();
  const int px = (int)rint((double)K.px), py = (int)rint((double)K.py);
  const double ori = K.ori;
  const double hist_width = 3.0 * (double)K.scl;
  int radius = (int)rint(hist_width * 1.4142135623730951 * (kD + 1) * 0.5);
  radius = min(radius, (int)sqrt((double)h * h + (double)w * w));
  const float cos_t = (float)(cos(ori * (M_PI / 180.0)) / hist_width), sin_t = (float)(sin(ori * (M_PI / 180.0)) / hist_width);
  const float fori = (float)ori;
  const int side = 2 * radius + 1;
  for (long k = lane; k < (long)side * side; k += 64) {
    const int di = (int)(k / side) - radius, dj = (int)(k % side) - radius;
    const float c_rot = (float)((double)dj * (double)cos_t - (double)di * (double)sin_t);
    const float r_rot = (float)((double)dj * (double)sin_t + (double)di * (double)cos_t);
    float rbin = r_rot + (kD / 2 - 0.5f), cbin = c_rot + (kD / 2 - 0.5f);
    const int r = py + di, c = px + dj;
    if (!(rbin > -1.f && rbin < (float)kD && cbin > -1.f && cbin < (float)kD && r > 0 && r < h - 1 && c > 0 && c < w - 1)) continue;
    const float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1];
    const float dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
    const float wgt = expf((c_rot * c_rot + r_rot * r_rot) * (float)(-1.0 / (kD * kD * 0.5)));
    float o = atan2f(dy, dx) * 57.29577951308232f;
    if (o < 0.f) o = o + 360.f;
    const float mag = sqrtf(dx * dx + dy * dy) * wgt;
    float obin = (o - fori) * (float)(kN / 360.0);
    const float r0f = floorf(rbin), c0f = floorf(cbin), o0f = floorf(obin);
    rbin = rbin - r0f; cbin = cbin - c0f; obin = obin - o0f;
    const int r0 = (int)r0f, c0 = (int)c0f;
    int o0 = (int)o0f;
    if (o0 < 0) o0 += kN;
    if (o0 >= kN) o0 -= kN;
    const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
    const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
    const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
    const float vv[4] = {v_rc00, v_rc01, v_rc10, v_rc11};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = ((r0 + 1 + (q >> 1)) * (kD + 2) + (c0 + 1 + (q & 1))) * (kN + 2) + o0;
      const float v1 = vv[q] * obin, v0 = vv[q] - v1;
      atomicAdd(&s_hist[wave][idx], (double)v0);
      atomicAdd(&s_hist[wave][idx + 1], (double)v1);
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // 128 outputs, two per lane; orientation bins n and n+1 wrap onto 0 and 1
  float v[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int k = lane + 64 * t, ob = k & 7, cell = k >> 3, ci = cell >> 2, cj = cell & 3;
    const int idx = ((ci + 1) * (kD + 2) + (cj + 1)) * (kN + 2);
    float x = (float)s_hist[wave][idx + ob];
    if (ob < 2) x = x + (float)s_hist[wave][idx + kN + ob];
    v[t] = x;
  }
  double ss = (double)v[0] * (double)v[0] + (double)v[1] * (double)v[1];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float thr = (float)sqrt(ss) * 0.2f;
  v[0] = fminf(v[0], thr); v[1] = fminf(v[1], thr);
  ss = (double)v[0] * (double)v[0] + (double)v[1] * (double)v[1];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float nrm = 512.0f / fmaxf((float)sqrt(ss), 1.19e-7f);
#pragma unroll
  for (int t = 0; t < 2; ++t) v[t] = fminf(fmaxf(rintf(v[t] * nrm), 0.f), 255.f);
  if (root) {
    double sum = (double)v[0] + (double)v[1];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float den = (float)sum + root_eps;
    v[0] = sqrtf(v[0] / den); v[1] = sqrtf(v[1] / den);
  }
  (void)s_dst;
  desc[(size_t)i * 128 + lane] = v[0];
  desc[(size_t)i * 128 + 64 + lane] = v[1];
}

Taps make_taps(double sigma) {
  const int ksize = ((int)std::nearbyint(sigma * 8 + 1)) | 1;
  const int r = ksize / 2;
  GTX_CHECK(r <= kMaxRadius, "sift: Gaussian radius %d exceeds %d", r, kMaxRadius);
  std::vector<double> k(2 * r + 1);
  double sum = 0;
  for (int i = -r; i <= r; ++i) { k[i + r] = std::exp(-((double)i * i) / (2.0 * sigma * sigma)); sum += k[i + r]; }
  Taps t{};
  t.r = r;
  for (int i = 0; i <= r; ++i) t.w[i] = (float)(k[r + i] / sum);
  return t;
}

}  // namespace

struct Sift::Impl {
  int device;
  hipStream_t s;
  int max_h, max_w;
  OctaveTable T{};
  DevBuf pyr, tmp, frame, gray, cand, refined, oriented, counters, finals, desc, xy;
  size_t cand_cap = 0, kp_cap = 0;
  int n = 0;
  std::vector<SiftKeypoint> host_kps;
  // GPU time of the stages of the last detect_and_compute (HIP events on the stream): 0 upload + gray + upscale + Gaussian / DoG pyramid,
  // 1 extrema + refine + orientation (with their host round trips for the counters), 2 descriptors; 3 = doubled-base pixels
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  float stage_ms[4] = {0, 0, 0, 0};
  ~Impl() {
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  }

  // dst = Gaussian(src, sigma); dog (may be null) = dst - src. GTX_SIFT_TWO_PASS=1: the two row / column passes through the scratch
  // image and a subtraction pass (the round-1 form, kept for the A/B and the bit-identity test between the two)
  void blur(const float* src, float* dst, float* dog, int w, int h, double sigma) {
    const Taps t = make_taps(sigma);
    static const bool two_pass = [] { const char* e = getenv("GTX_SIFT_TWO_PASS"); return e && e[0] == '1'; }();
    if (two_pass) {
      hipLaunchKernelGGL(blur_h_kernel, dim3(cdiv(w, 256), h), dim3(256), 0, s, src, tmp.as<float>(), w, h, t);
      hipLaunchKernelGGL(blur_v_kernel, dim3(cdiv(w, 256), h), dim3(256), 0, s, tmp.as<float>(), dst, w, h, t);
      const size_t np = (size_t)w * h;
      if (dog) hipLaunchKernelGGL(sub_kernel, dim3((unsigned)cdiv((long)np, 256L)), dim3(256), 0, s, dst, src, dog, np);
      return;
    }
    const dim3 grid(cdiv(w, kBlurTW), cdiv(h, kBlurTH));
    switch (t.r) {                                           // the default scale space's radii at compile time; anything else: the generic tile
      case 5: hipLaunchKernelGGL(blur_tile_kernel_r<5>, grid, dim3(256), 0, s, src, dst, dog, w, h, t); break;
      case 6: hipLaunchKernelGGL(blur_tile_kernel_r<6>, grid, dim3(256), 0, s, src, dst, dog, w, h, t); break;
      case 8: hipLaunchKernelGGL(blur_tile_kernel_r<8>, grid, dim3(256), 0, s, src, dst, dog, w, h, t); break;
      case 10: hipLaunchKernelGGL(blur_tile_kernel_r<10>, grid, dim3(256), 0, s, src, dst, dog, w, h, t); break;
      case 13: hipLaunchKernelGGL(blur_tile_kernel_r<13>, grid, dim3(256), 0, s, src, dst, dog, w, h, t); break;
      default: hipLaunchKernelGGL(blur_tile_kernel, grid, dim3(256), 0, s, src, dst, dog, w, h, t);
    }
  }
};

Sift::Sift(int device, hipStream_t stream, int max_h, int max_w) : impl_(new Impl) {
  Impl& S = *impl_;
  S.device = device; S.s = stream; S.max_h = max_h; S.max_w = max_w;
  GTX_CHECK(max_h >= 8 && max_w >= 8, "sift: image %dx%d too small", max_w, max_h);
  GTX_CHECK(2 * max_h <= 65535, "sift: images taller than 32767 rows are not supported (grid.y)");
  GTX_HIP(hipSetDevice(device));
  const size_t base = (size_t)(2 * max_h) * (2 * max_w);
  size_t total = 0, px = base;
  int w = 2 * max_w, h = 2 * max_h;
  for (int o = 0; o < kMaxOct && w >= 1 && h >= 1; ++o) { total += (size_t)w * h * (kGauss + kDog); w /= 2; h /= 2; (void)px; }
  S.pyr.alloc(total * sizeof(float));
  S.tmp.alloc(base * sizeof(float));
  S.frame.alloc((size_t)max_h * max_w * 3);
  S.gray.alloc((size_t)max_h * max_w * sizeof(float));
  S.cand_cap = std::max<size_t>(1 << 16, base / 16);
  S.kp_cap = S.cand_cap;
  S.cand.alloc(S.cand_cap * sizeof(Cand));
  S.refined.alloc(S.cand_cap * sizeof(Refined));
  S.oriented.alloc(S.kp_cap * sizeof(Oriented));
  S.counters.alloc(4 * sizeof(int));
  for (hipEvent_t& e : S.ev) GTX_HIP(hipEventCreate(&e));
}

void Sift::stage_ms(float out[4]) const {
  for (int i = 0; i < 4; ++i) out[i] = impl_->stage_ms[i];
}

Sift::~Sift() = default;

int Sift::count() const { return impl_->n; }
const float* Sift::descriptors_dev() const { return impl_->desc.as<float>(); }
const float2* Sift::positions_dev() const { return impl_->xy.as<float2>(); }
int Sift::n_octaves() const { return impl_->T.n; }

void Sift::detect_and_compute(const uint8_t* image, int h, int w, int max_features, bool root, float root_eps) {
  Impl& S = *impl_;
  GTX_CHECK(image && h >= 8 && w >= 8 && h <= S.max_h && w <= S.max_w, "sift: image %dx%d outside [8, %dx%d]", w, h, S.max_w, S.max_h);
  GTX_CHECK(max_features >= 1, "sift: max_features must be positive");
  GTX_HIP(hipSetDevice(S.device));
  hipStream_t s = S.s;
  // ---- pyramid layout for this image
  const int bw = 2 * w, bh = 2 * h;
  OctaveTable& T = S.T;
  T.n = std::min(kMaxOct, (int)std::nearbyint(std::log((double)std::min(bw, bh)) / std::log(2.0) - 2) + 1);
  GTX_CHECK(T.n >= 1, "sift: image too small for one octave");
  float* p = S.pyr.as<float>();
  int ow = bw, oh = bh, n_oct = 0;
  for (int o = 0; o < T.n; ++o) {
    T.w[o] = ow; T.h[o] = oh;
    for (int i = 0; i < kGauss; ++i) { T.g[o][i] = p; p += (size_t)ow * oh; }
    for (int i = 0; i < kDog; ++i) { T.d[o][i] = p; p += (size_t)ow * oh; }
    n_oct = o + 1;
    if (std::min(ow, oh) / 2 < 1) break;
    ow /= 2; oh /= 2;
  }
  T.n = n_oct;
  // ---- base image
  GTX_HIP(hipEventRecord(S.ev[0], s));
  GTX_HIP(hipMemcpyAsync(S.frame.p, image, (size_t)h * w * 3, hipMemcpyHostToDevice, s));
  const size_t npx = (size_t)h * w;
  hipLaunchKernelGGL(gray_kernel, dim3((unsigned)cdiv((long)npx, 256L)), dim3(256), 0, s, S.frame.as<uint8_t>(), S.gray.as<float>(), npx);
  hipLaunchKernelGGL(upscale_kernel, dim3(cdiv(bw, 256), bh), dim3(256), 0, s, S.gray.as<float>(), h, w, T.g[0][1]);  // scratch: layer 1
  S.blur(T.g[0][1], T.g[0][0], nullptr, bw, bh, std::sqrt(std::max(kSigma * kSigma - 4 * 0.5 * 0.5, 0.01)));
  double sig[kGauss];
  {
    const double k = std::pow(2.0, 1.0 / kLayers);
    sig[0] = kSigma;
    for (int i = 1; i < kGauss; ++i) {
      const double prev = kSigma * std::pow(k, i - 1);
      sig[i] = std::sqrt((prev * k) * (prev * k) - prev * prev);
    }
  }
  for (int o = 0; o < T.n; ++o) {
    const int ww = T.w[o], hh = T.h[o];
    if (o > 0) {
      const int sw = T.w[o - 1], sh = T.h[o - 1];
      hipLaunchKernelGGL(down_kernel, dim3(cdiv(ww, 256), hh), dim3(256), 0, s, T.g[o - 1][kLayers], sh, sw, T.g[o][0], hh, ww,
                         (double)sh / hh, (double)sw / ww);
    }
    for (int i = 1; i < kGauss; ++i) S.blur(T.g[o][i - 1], T.g[o][i], T.d[o][i - 1], ww, hh, sig[i]);   // DoG layer i - 1 = g[i] - g[i - 1], written by the blur
  }
  GTX_HIP(hipEventRecord(S.ev[1], s));
  // ---- extrema -> refine -> orientation
  int* cnt = S.counters.as<int>();
  GTX_HIP(hipMemsetAsync(cnt, 0, 4 * sizeof(int), s));
  const float thr = (float)std::floor(0.5 * kContrastThr / kLayers * 255);
  for (int o = 0; o < T.n; ++o) {
    const int ww = T.w[o], hh = T.h[o];
    if (ww <= 2 * kBorder || hh <= 2 * kBorder) continue;
    for (int i = 1; i <= kLayers; ++i)
      hipLaunchKernelGGL(extrema_kernel, dim3(cdiv(ww - 2 * kBorder, 256), hh - 2 * kBorder), dim3(256), 0, s, T.d[o][i - 1], T.d[o][i],
                         T.d[o][i + 1], ww, hh, thr, o, i, S.cand.as<Cand>(), cnt, (int)S.cand_cap);
  }
  int hc[4];
  GTX_HIP(hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
  const int n_cand = std::min<long>(hc[0], (long)S.cand_cap);
  if (n_cand > 0) hipLaunchKernelGGL(refine_kernel, dim3(cdiv(n_cand, 256)), dim3(256), 0, s, T, S.cand.as<Cand>(), n_cand, S.refined.as<Refined>(), cnt + 1);
  GTX_HIP(hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
  const int n_ref = hc[1];
  if (n_ref > 0)
    hipLaunchKernelGGL(orient_kernel, dim3(cdiv(n_ref, 4)), dim3(256), 0, s, T, S.refined.as<Refined>(), n_ref, S.oriented.as<Oriented>(), cnt + 2,
                       (int)S.kp_cap);
  GTX_HIP(hipMemcpyAsync(hc, cnt, sizeof hc, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
  GTX_HIP(hipEventRecord(S.ev[2], s));
  const int n_ori = std::min<long>(hc[2], (long)S.kp_cap);
  std::vector<Oriented> ori(n_ori);
  if (n_ori) GTX_HIP(hipMemcpy(ori.data(), S.oriented.p, sizeof(Oriented) * n_ori, hipMemcpyDeviceToHost));
  // OpenCV order: octave, layer, row, column of the scale-space extremum, then orientation bin -- one 64-bit key per keypoint.
  // retainBest (more keypoints than max_features: the strongest responses stay, in that order; equal responses: the earlier key
  // first, which is what a stable sort by response of the ordered list keeps) is a selection, not a sort: nth_element on
  // (response desc, key asc), then only the kept ones are ordered by key. At the reference's size (1.4 M oriented keypoints of a
  // 15 000-px cut-out, 250 000 kept) the two full sorts of the structs this replaces took 60 of the stage's 73 ms.
  {
    std::vector<uint64_t> key(n_ori);
    parallel_for(8, [&](int part) {
      for (size_t i = (size_t)n_ori * part / 8; i < (size_t)n_ori * (part + 1) / 8; ++i) {
        const Oriented& a = ori[i];
        key[i] = ((uint64_t)a.key.o << 56) | ((uint64_t)a.key.layer << 48) | ((uint64_t)(uint32_t)a.key.r << 28) | ((uint64_t)(uint32_t)a.key.c << 8) |
                 (uint64_t)(uint32_t)a.bin;            // o < 16, layer < 8, r and c < 2^20, bin < 256
      }
    });
    std::vector<uint32_t> idx(n_ori);
    for (uint32_t i = 0; i < (uint32_t)n_ori; ++i) idx[i] = i;
    size_t keep = idx.size();
    if ((int)idx.size() > max_features) {
      keep = (size_t)max_features;
      std::nth_element(idx.begin(), idx.begin() + (long)keep, idx.end(), [&](uint32_t a, uint32_t b) {
        return ori[a].response != ori[b].response ? ori[a].response > ori[b].response : key[a] < key[b];
      });
      idx.resize(keep);
    }
    std::vector<std::pair<uint64_t, uint32_t>> order(keep);
    for (size_t i = 0; i < keep; ++i) order[i] = {key[idx[i]], idx[i]};
    std::sort(order.begin(), order.end());
    std::vector<Oriented> kept(keep);
    for (size_t i = 0; i < keep; ++i) kept[i] = ori[order[i].second];
    ori.swap(kept);
  }
  const int n = (int)ori.size();
  S.n = n;
  S.host_kps.resize(n);
  std::vector<Final> fin(n);
  std::vector<float2> xy(n);
  for (int i = 0; i < n; ++i) {
    const Oriented& O = ori[i];
    const double scale = 1.0 / (double)(1 << O.o);
    Final f;
    f.px = (float)((double)O.x * scale); f.py = (float)((double)O.y * scale);
    double a = 360.0 - (double)O.angle;
    if (std::fabs(a - 360.0) < 1.19e-7) a = 0.0;
    f.ori = a;
    f.scl = (float)((double)O.size * scale * 0.5);
    f.o = O.o; f.layer = O.layer;
    fin[i] = f;
    SiftKeypoint k;
    k.x = (float)((double)O.x * 0.5); k.y = (float)((double)O.y * 0.5); k.size = (float)((double)O.size * 0.5);
    k.angle = O.angle; k.response = O.response;
    k.octave = (O.word & ~255) | ((O.o - 1) & 255);
    S.host_kps[i] = k;
    xy[i] = make_float2(k.x, k.y);
  }
  if (S.finals.bytes < sizeof(Final) * std::max(n, 1)) S.finals.alloc(sizeof(Final) * std::max(n, 1) * 2);
  if (S.desc.bytes < sizeof(float) * 128 * std::max(n, 1)) S.desc.alloc(sizeof(float) * 128 * std::max(n, 1) * 2);
  if (S.xy.bytes < sizeof(float2) * std::max(n, 1)) S.xy.alloc(sizeof(float2) * std::max(n, 1) * 2);
  if (n > 0) {
    GTX_HIP(hipMemcpyAsync(S.finals.p, fin.data(), sizeof(Final) * n, hipMemcpyHostToDevice, s));
    GTX_HIP(hipMemcpyAsync(S.xy.p, xy.data(), sizeof(float2) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(describe_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, T, S.finals.as<Final>(), n, S.desc.as<float>(), root ? 1 : 0, root_eps);
    GTX_HIP(hipGetLastError());
  }
  GTX_HIP(hipEventRecord(S.ev[3], s));
  GTX_HIP(hipStreamSynchronize(s));     // fin / xy are stack-owned host buffers
  for (int i = 0; i < 3; ++i) GTX_HIP(hipEventElapsedTime(&S.stage_ms[i], S.ev[i], S.ev[i + 1]));
  S.stage_ms[3] = (float)((double)bw * bh);
}

const std::vector<SiftKeypoint>& Sift::keypoints_host() const { return impl_->host_kps; }

void Sift::download(std::vector<SiftKeypoint>& kps, std::vector<float>& desc) const {
  const Impl& S = *impl_;
  kps = S.host_kps;
  desc.resize((size_t)S.n * 128);
  if (S.n) {
    GTX_HIP(hipSetDevice(S.device));
    GTX_HIP(hipStreamSynchronize(S.s));
    GTX_HIP(hipMemcpy(desc.data(), S.desc.p, desc.size() * sizeof(float), hipMemcpyDeviceToHost));
  }
}

void Sift::pyramid_image(int kind, int octave, int layer, std::vector<float>& out, int* h, int* w) const {
  const Impl& S = *impl_;
  GTX_CHECK(octave >= 0 && octave < S.T.n && layer >= 0 && layer < (kind == 0 ? kGauss : kDog), "sift: no pyramid image (%d, %d, %d)", kind, octave, layer);
  *h = S.T.h[octave]; *w = S.T.w[octave];
  out.resize((size_t)*h * *w);
  GTX_HIP(hipSetDevice(S.device));
  GTX_HIP(hipStreamSynchronize(S.s));
  GTX_HIP(hipMemcpy(out.data(), kind == 0 ? S.T.g[octave][layer] : S.T.d[octave][layer], out.size() * sizeof(float), hipMemcpyDeviceToHost));
}

}  // namespace gtx
