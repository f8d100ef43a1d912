// BoT-SORT global motion compensation, method 'sparseOptFlow', on the GPU (gfx950).
// Replaces ultralytics.trackers.utils.gmc.GMC.apply_sparseoptflow as the tracker callback of
// model.track(..., persist=True) runs it for tracker_type 'botsort' (reference: geotrax/extract.py:153,
// geotrax/cfg/default.yaml:362-374): Shi-Tomasi corners of the half
// resolution gray frame (<= 1000, quality 0.01), pyramidal Lucas-Kanade against the previous frame
// (21x21 window, 3 levels above the base, 30 iterations / 0.01), RANSAC similarity transform.
// oracle/gmc_ref.py is the line-by-line CPU restatement.
//
// Kernels (all on the context's stream, one submit = ~10 launches, no host round trip):
//   response  : integer Sobel 3x3 + 3x3 box sums -> min eigenvalue (f64) per pixel + global max
//   nms       : threshold at 0.01 max, 3x3 local maxima -> candidate list
//   select    : one workgroup: radix select of the 1000 strongest (ties: larger pixel index),
//               bitonic sort in LDS -> corner list, strongest first
//   pyrdown   : 5x5 [1 4 6 4 1] integer pyramid, 3 levels
//   lk        : one wave per corner; the 22x22 neighbourhood of the previous image, its Scharr
//               derivatives and (per iteration) of the current image live in LDS; the 21x21 sums are
//               wave reductions in f64
//   compact   : ordered compaction of the tracked pairs
//   ransac    : one wave per two-point hypothesis, inlier count; argmax (first best)
// collect(): least squares on the inliers (3 rounds) on the host, translation scaled back by 2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "gmc.hpp"

namespace gtx {

namespace {

constexpr int kMaxCorners = 1000, kWin = 21, kMaxLevel = 3, kMaxIters = 30, kHyp = 512;
constexpr double kQuality = 0.01, kEps = 0.01, kMinEig = 1e-4, kRansacThr = 3.0;
constexpr int kPatch = kWin + 1;   // 22: samples needed for bilinear interpolation of a 21-wide window

struct Cand { double val; int pix; int pad; };

struct Pyr {
  const uint8_t* img[kMaxLevel + 1];
  int w[kMaxLevel + 1], h[kMaxLevel + 1];
};

struct GmcResult {
  int n_prev, n_valid, best_count, pad;
  double a, b, tx, ty;
};

__device__ __forceinline__ int refl101(int i, int n) {
  const int p = 2 * (n - 1);
  i = (i < 0 ? -i : i) % p;
  return i >= n ? p - i : i;
}

// BGR u8 -> gray (cv2 fixed point) -> exact 2x2 mean, as the detector's preprocess pass writes it
__global__ __launch_bounds__(256) void gray_half_kernel(const uint8_t* __restrict__ bgr, int w, uint8_t* __restrict__ out, int oh, int ow) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const uint8_t* r0 = bgr + ((size_t)(2 * y) * w + 2 * x) * 3;
  const uint8_t* r1 = r0 + (size_t)w * 3;
  auto gr = [](const uint8_t* p) { return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14; };
  out[(size_t)y * ow + x] = (uint8_t)((gr(r0) + gr(r0 + 3) + gr(r1) + gr(r1 + 3) + 2) >> 2);
}

// ---- corners
__device__ __forceinline__ void sobel_at(const uint8_t* __restrict__ g, int w, int h, int y, int x, long& dx, long& dy) {
  const int ym = refl101(y - 1, h), yp = refl101(y + 1, h), xm = refl101(x - 1, w), xp = refl101(x + 1, w);
  const int a = g[(size_t)ym * w + xm], b = g[(size_t)ym * w + x], c = g[(size_t)ym * w + xp];
  const int d = g[(size_t)y * w + xm], f = g[(size_t)y * w + xp];
  const int k = g[(size_t)yp * w + xm], l = g[(size_t)yp * w + x], m = g[(size_t)yp * w + xp];
  dx = (c - a) + 2 * (f - d) + (m - k);
  dy = (k - a) + 2 * (l - b) + (m - c);
}

// reflect-101 for indices at most a few pixels outside [0, n)
__device__ __forceinline__ int refl_near(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// 64x16 output tile: gray (+2 ring) -> LDS, Sobel at the (+1 ring) positions -> LDS, 3x3 box sums.
// All sums fit int32 (|d| <= 1020, 9 d^2 <= 9.4e6); the eigenvalue is f64.
constexpr int kRT_W = 64, kRT_H = 16;
__global__ __launch_bounds__(256) void response_kernel(const uint8_t* __restrict__ g, int w, int h, double* __restrict__ lam,
                                                       unsigned long long* __restrict__ max_bits) {
  __shared__ uint8_t s_g[(kRT_H + 4) * (kRT_W + 4)];
  __shared__ short s_dx[(kRT_H + 2) * (kRT_W + 2)], s_dy[(kRT_H + 2) * (kRT_W + 2)];
  const int x0 = blockIdx.x * kRT_W, y0 = blockIdx.y * kRT_H;
  // The box sum at (y, x) uses Sobel values at reflect(y+i), reflect(x+j); the Sobel there uses gray at
  // reflect(. +-1) of THAT position. Away from the image border both are plain offsets; at the border the
  // tile is filled per Sobel position, so stage the Sobel values directly.
  for (int i = threadIdx.x; i < (kRT_H + 4) * (kRT_W + 4); i += 256) {
    const int ty = i / (kRT_W + 4), tx = i % (kRT_W + 4);
    s_g[i] = g[(size_t)refl_near(min(y0 + ty - 2, h + 1), h) * w + refl_near(min(x0 + tx - 2, w + 1), w)];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (kRT_H + 2) * (kRT_W + 2); i += 256) {
    const int ty = i / (kRT_W + 2), tx = i % (kRT_W + 2);
    // Sobel position in image coordinates (may be one outside: reflect it, then its neighbours reflect again)
    const int py = refl_near(min(y0 + ty - 1, h), h), px = refl_near(min(x0 + tx - 1, w), w);
    int dx, dy;
    const bool interior = py >= 1 && py < h - 1 && px >= 1 && px < w - 1 && py == y0 + ty - 1 && px == x0 + tx - 1;
    if (interior) {
      const uint8_t* c = s_g + (ty + 1) * (kRT_W + 4) + tx + 1;
      constexpr int S = kRT_W + 4;
      dx = (c[-S + 1] - c[-S - 1]) + 2 * (c[1] - c[-1]) + (c[S + 1] - c[S - 1]);
      dy = (c[S - 1] - c[-S - 1]) + 2 * (c[S] - c[-S]) + (c[S + 1] - c[-S + 1]);
    } else {
      long ldx, ldy;
      sobel_at(g, w, h, py, px, ldx, ldy);
      dx = (int)ldx; dy = (int)ldy;
    }
    s_dx[i] = (short)dx; s_dy[i] = (short)dy;
  }
  __syncthreads();
  double m = 0.0;
  for (int i = threadIdx.x; i < kRT_H * kRT_W; i += 256) {
    const int ty = i / kRT_W, tx = i % kRT_W;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= w || y >= h) continue;
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
      for (int dj = 0; dj < 3; ++dj) {
        const int q = (ty + di) * (kRT_W + 2) + tx + dj;
        const int dx = s_dx[q], dy = s_dy[q];
        a += dx * dx; b += dx * dy; c += dy * dy;
      }
    const double v = 0.5 * (double)(a + c) - sqrt(0.25 * ((double)(a - c) * (double)(a - c)) + (double)b * (double)b);
    lam[(size_t)y * w + x] = v;
    m = fmax(m, v);
  }
  // block max -> global max (bit pattern of a non-negative double is order preserving). Thousands of
  // atomics on one address serialise in L2: reduce to one per workgroup and skip it when the value already
  // published is at least as large (a plain read is enough for that: the maximum only grows).
  __shared__ double s_m[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmax(fmax(s_m[0], s_m[1]), fmax(s_m[2], s_m[3]));
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    if (m > 0.0 && bits > __atomic_load_n(max_bits, __ATOMIC_RELAXED)) atomicMax(max_bits, bits);
  }
}

// A workgroup takes 256 columns x kNmsRows rows and appends its corners to an LDS list first: one reservation on the global
// counter per WORKGROUP. With one per wave (32 k waves of a 1920 x 1080 map, nearly all of them holding a corner) the
// same-address atomics queue up in L2 at ~1.7 ns each and were the kernel's whole 58 us.
constexpr int kNmsRows = 8, kNmsList = 1024;
__global__ __launch_bounds__(256) void nms_kernel(const double* __restrict__ lam, int w, int h, const unsigned long long* __restrict__ max_bits,
                                                  Cand* __restrict__ cand, int* __restrict__ n_cand, int cap, int* __restrict__ hist16) {
  __shared__ Cand s_list[kNmsList];
  __shared__ int s_hist[256];
  __shared__ int s_n, s_base;
  s_hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int x = 1 + blockIdx.x * 256 + threadIdx.x;
  const unsigned long long mb = *max_bits;
  const double thr = __longlong_as_double((long long)mb) * kQuality;
  // bucket = distance of the key's top 16 bits (exponent + 4 mantissa bits) below the maximum's: the
  // candidates span a factor 100 (quality 0.01), i.e. ~107 of the 256 buckets
  auto bucket = [mb](double val) { return min(max((int)(mb >> 48) - (int)((unsigned long long)__double_as_longlong(val) >> 48), 0), 255); };
  for (int r = 0; r < kNmsRows; ++r) {
    const int y = 1 + blockIdx.y * kNmsRows + r;
    bool keep = x < w - 1 && y < h - 1;
    double v = 0.0;
    if (keep) {
      v = lam[(size_t)y * w + x];
      keep = v > thr;
    }
    if (keep) {
      double mx = v;
#pragma unroll
      for (int i = -1; i <= 1; ++i)
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
          const double q = lam[(size_t)(y + i) * w + x + j];
          mx = fmax(mx, q > thr ? q : 0.0);
        }
      keep = v == mx;
    }
    const int slot = gtx_wave_append(&s_n, keep);           // LDS counter; one LDS atomic per wave
    if (keep) {
      const Cand c{v, y * w + x, 0};
      if (slot < kNmsList) {
        s_list[slot] = c;
      } else {                                              // a plateau of equal maxima overflowing the list: straight to the global one
        const int g = atomicAdd(n_cand, 1);
        if (g < cap) {
          cand[g] = c;
          atomicAdd(&s_hist[bucket(v)], 1);
        }
      }
    }
  }
  __syncthreads();
  const int nloc = min(s_n, kNmsList);
  if (threadIdx.x == 0 && nloc > 0) s_base = atomicAdd(n_cand, nloc);
  __syncthreads();
  if (nloc > 0) {
    const int base = s_base;
    for (int k = threadIdx.x; k < nloc; k += 256)
      if (base + k < cap) {                                 // a candidate that is stored is counted, as before
        cand[base + k] = s_list[k];
        atomicAdd(&s_hist[bucket(s_list[k].val)], 1);
      }
  }
  __syncthreads();
  if (s_hist[threadIdx.x] != 0) atomicAdd(&hist16[threadIdx.x], s_hist[threadIdx.x]);
}

// top-kMaxCorners by (val desc, pix desc), sorted; one workgroup of 1024 threads.
// The candidates' histogram over the top 16 key bits (filled by nms_kernel) gives the bucket the 1000th
// strongest falls into; one sweep then gathers everything above that bucket plus the bucket itself
// (a 6 %-wide value range: a few hundred candidates) into LDS, where a bitonic sort of (key, pix)
// finishes the job. A bucket too full for LDS is narrowed by further 8-bit radix passes first.
constexpr int kSelCap = 4096;
__global__ __launch_bounds__(1024) void select_kernel(const Cand* __restrict__ cand, int* __restrict__ n_cand, int cap, int w,
                                                      int* __restrict__ hist16, unsigned long long* __restrict__ max_bits_p,
                                                      float2* __restrict__ pts, int* __restrict__ n_pts) {
  __shared__ unsigned long long s_key[kSelCap];
  __shared__ int s_pix[kSelCap];
  __shared__ int s_hist[256];
  __shared__ unsigned long long s_lo;       // keys >= s_lo are gathered
  __shared__ int s_cnt, s_bucket, s_above, s_need;
  const int tid = threadIdx.x;
  const int n = min(*n_cand, cap);
  const int want = min(n, kMaxCorners);
  if (tid == 0) { s_lo = 0; s_cnt = 0; }
  __syncthreads();
  if (n > kSelCap) {
    // bucket of the want-th strongest: the 256 relative buckets, strongest first
    if (tid == 0) {
      int acc = 0, bkt = 0;
      for (; bkt < 255; ++bkt) { const int c = hist16[bkt]; if (acc + c >= want) break; acc += c; }
      s_bucket = (int)(*max_bits_p >> 48) - bkt;              // back to the absolute top-16-bit value
      s_above = acc; s_need = want - acc; s_cnt = hist16[bkt];
    }
    __syncthreads();
    unsigned long long lo = (unsigned long long)s_bucket << 48;
    int population = s_cnt;
    __syncthreads();
    // narrow an over-full bucket: 8 more key bits per pass, keeping the part that still holds the cut
    for (int shift = 40; population + s_above > kSelCap - 64 && shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      const unsigned long long mask = ~0ull << (shift + 8);
      for (int i = tid; i < n; i += 1024) {
        const unsigned long long k = (unsigned long long)__double_as_longlong(cand[i].val);
        if ((k & mask) == lo) atomicAdd(&s_hist[(k >> shift) & 255], 1);
      }
      __syncthreads();
      if (tid == 0) {
        int need = s_need, b2 = 255;
        for (; b2 > 0; --b2) { if (s_hist[b2] >= need) break; need -= s_hist[b2]; s_above += s_hist[b2]; }
        s_need = need; s_bucket = b2; s_cnt = s_hist[b2];
      }
      __syncthreads();
      lo |= (unsigned long long)s_bucket << shift;
      population = s_cnt;
      __syncthreads();
    }
    if (tid == 0) { s_lo = lo; s_cnt = 0; }
    __syncthreads();
  }
  for (int i = tid; i < kSelCap; i += 1024) { s_key[i] = 0; s_pix[i] = -1; }
  __syncthreads();
  const unsigned long long lo = s_lo;
  for (int i = tid; i < n; i += 1024) {
    const unsigned long long k = (unsigned long long)__double_as_longlong(cand[i].val);
    if (k >= lo) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < kSelCap) { s_key[slot] = k; s_pix[slot] = cand[i].pix; }
    }
  }
  __syncthreads();
  // bitonic sort of the occupied slots (padded to a power of two), descending by (key, pix); empty
  // slots (key 0, pix -1) sink to the end
  int m = 1024;
  while (m < min(s_cnt, kSelCap)) m <<= 1;
  for (int k2 = 2; k2 <= m; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < m; t += 1024) {
        const int ixj = t ^ j;
        if (ixj > t) {
          const unsigned long long ka = s_key[t], kb = s_key[ixj];
          const int pa = s_pix[t], pb = s_pix[ixj];
          const bool a_first = ka > kb || (ka == kb && pa > pb);
          const bool desc = (t & k2) == 0;
          if (desc ? !a_first : a_first) { s_key[t] = kb; s_key[ixj] = ka; s_pix[t] = pb; s_pix[ixj] = pa; }
        }
      }
      __syncthreads();
    }
  if (tid < want) pts[tid] = make_float2((float)(s_pix[tid] % w), (float)(s_pix[tid] / w));
  if (tid == 0) *n_pts = want;
  // last reader of the frame's counters: clear them for the next frame of this parity (two memset launches less per
  // frame on the tracker's critical path; they start out zero)
  if (tid < 256) hist16[tid] = 0;
  if (tid == 0) { *max_bits_p = 0ull; *n_cand = 0; }
}

// ---- pyramid
__global__ __launch_bounds__(256) void pyrdown_kernel(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= dw) return;
  const int k[5] = {1, 4, 6, 4, 1};
  int acc = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const uint8_t* row = src + (size_t)refl101(2 * y + i - 2, sh) * sw;
    int r = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) r += k[j] * row[refl101(2 * x + j - 2, sw)];
    acc += k[i] * r;
  }
  dst[(size_t)y * dw + x] = (uint8_t)((acc + 128) >> 8);
}

// ---- Lucas-Kanade: one wave per point
constexpr int kWinSamples = kWin * kWin;            // 441
constexpr int kPerLane = (kWinSamples + 63) / 64;   // 7

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void lk_kernel(const Pyr P, const Pyr C, const float2* __restrict__ pts, const int* __restrict__ n_pts,
                                                 float2* __restrict__ out, int* __restrict__ status) {
  // Patches are kept in LDS in the narrowest exact form -- pixels as bytes, the Scharr sums as their integer numerators
  // (|3(c-a) + 10(f-d) + 3(m-g)| <= 4080) -- and widened to double when read: the same doubles as before, bit for bit, in
  // 11.6 KB per workgroup instead of 62 KB (at 62 KB the 250 workgroups of a frame took a third of every CU's LDS away from
  // the convolutions that share the GPU).
  __shared__ uint8_t s_I[4][kPatch * kPatch], s_J[4][kPatch * kPatch];
  __shared__ short s_Ix[4][kPatch * kPatch], s_Iy[4][kPatch * kPatch];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= *n_pts) return;
  const double px = (double)pts[n].x, py = (double)pts[n].y;
  const double half = (kWin - 1) * 0.5;
  double nx = 0.0, ny = 0.0;
  bool ok = true;
  constexpr int kPatchSlots = (kPatch * kPatch + 63) / 64;
  int prow[kPatchSlots], pcol[kPatchSlots];        // this lane's patch positions (the same at every level and iteration)
#pragma unroll
  for (int t = 0; t < kPatchSlots; ++t) { prow[t] = (lane + 64 * t) / kPatch; pcol[t] = (lane + 64 * t) % kPatch; }
  for (int L = kMaxLevel; L >= 0; --L) {
    const double sc = 1.0 / (double)(1 << L);
    const double qx = px * sc, qy = py * sc;
    if (L == kMaxLevel) { nx = qx; ny = qy; } else { nx = nx * 2.0; ny = ny * 2.0; }
    const uint8_t* I = P.img[L];
    const uint8_t* J = C.img[L];
    const int w = P.w[L], h = P.h[L];
    const double tx = qx - half, ty = qy - half;
    const int x0 = (int)floor(tx), y0 = (int)floor(ty);
    if (x0 < 0 || y0 < 0 || x0 + kWin + 1 > w || y0 + kWin + 1 > h) {
      if (L == 0) ok = false;
      continue;
    }
    const double fx = tx - x0, fy = ty - y0;
    // previous image patch and its Scharr derivatives at the 22x22 integer positions
#pragma unroll
    for (int t = 0; t < kPatchSlots; ++t) {
      const int k = lane + 64 * t;
      if (k >= kPatch * kPatch) break;
      const int yy = y0 + prow[t], xx = x0 + pcol[t];
      const int ym = refl101(yy - 1, h), yp = refl101(yy + 1, h), xm = refl101(xx - 1, w), xp = refl101(xx + 1, w);
      const int a = I[(size_t)ym * w + xm], b = I[(size_t)ym * w + xx], c = I[(size_t)ym * w + xp];
      const int d = I[(size_t)yy * w + xm], f = I[(size_t)yy * w + xp];
      const int g = I[(size_t)yp * w + xm], hh = I[(size_t)yp * w + xx], m = I[(size_t)yp * w + xp];
      s_I[wave][k] = I[(size_t)yy * w + xx];
      s_Ix[wave][k] = (short)(3 * (c - a) + 10 * (f - d) + 3 * (m - g));      // x 1/32 when read
      s_Iy[wave][k] = (short)(3 * (g - a) + 10 * (hh - b) + 3 * (m - c));
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const double w00 = (1 - fx) * (1 - fy), w01 = fx * (1 - fy), w10 = (1 - fx) * fy, w11 = fx * fy;
    double Iw[kPerLane], Ixw[kPerLane], Iyw[kPerLane];
    double a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
    for (int t = 0; t < kPerLane; ++t) {
      const int k = lane + 64 * t;
      Iw[t] = 0; Ixw[t] = 0; Iyw[t] = 0;
      if (k < kWinSamples) {
        const int r = k / kWin, c = k % kWin, q = r * kPatch + c;
        auto dx_ = [&](int i) { return (double)s_Ix[wave][i] / 32.0; };
        auto dy_ = [&](int i) { return (double)s_Iy[wave][i] / 32.0; };
        Iw[t] = (double)s_I[wave][q] * w00 + (double)s_I[wave][q + 1] * w01 + (double)s_I[wave][q + kPatch] * w10 + (double)s_I[wave][q + kPatch + 1] * w11;
        Ixw[t] = dx_(q) * w00 + dx_(q + 1) * w01 + dx_(q + kPatch) * w10 + dx_(q + kPatch + 1) * w11;
        Iyw[t] = dy_(q) * w00 + dy_(q + 1) * w01 + dy_(q + kPatch) * w10 + dy_(q + kPatch + 1) * w11;
        a11 += Ixw[t] * Ixw[t]; a12 += Ixw[t] * Iyw[t]; a22 += Iyw[t] * Iyw[t];
      }
    }
    const double A11 = wave_sum(a11), A12 = wave_sum(a12), A22 = wave_sum(a22);
    double D = A11 * A22 - A12 * A12;
    const double min_eig = (A22 + A11 - sqrt((A11 - A22) * (A11 - A22) + 4.0 * A12 * A12)) / (2.0 * kWin * kWin);
    if (min_eig < kMinEig || D < 1.1920929e-7) {
      if (L == 0) ok = false;
      continue;
    }
    D = 1.0 / D;
    double pdx = 0.0, pdy = 0.0;
    int last_jx0 = -(1 << 30), last_jy0 = 0;
    for (int j = 0; j < kMaxIters; ++j) {
      const double ux = nx - half, uy = ny - half;
      const int jx0 = (int)floor(ux), jy0 = (int)floor(uy);
      if (jx0 < 0 || jy0 < 0 || jx0 + kWin + 1 > w || jy0 + kWin + 1 > h) {
        if (L == 0) ok = false;
        break;
      }
      const double gx = ux - jx0, gy = uy - jy0;
      if (jx0 != last_jx0 || jy0 != last_jy0) {     // sub-pixel steps mostly stay inside the same integer window: the staged patch is still it
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < kPatchSlots; ++t) {
          const int k = lane + 64 * t;
          if (k < kPatch * kPatch) s_J[wave][k] = J[(size_t)(jy0 + prow[t]) * w + jx0 + pcol[t]];
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        last_jx0 = jx0; last_jy0 = jy0;
      }
      const double v00 = (1 - gx) * (1 - gy), v01 = gx * (1 - gy), v10 = (1 - gx) * gy, v11 = gx * gy;
      double b1 = 0, b2 = 0;
#pragma unroll
      for (int t = 0; t < kPerLane; ++t) {
        const int k = lane + 64 * t;
        if (k < kWinSamples) {
          const int r = k / kWin, c = k % kWin, q = r * kPatch + c;
          const double Jw = (double)s_J[wave][q] * v00 + (double)s_J[wave][q + 1] * v01 + (double)s_J[wave][q + kPatch] * v10 +
                            (double)s_J[wave][q + kPatch + 1] * v11;
          const double diff = Jw - Iw[t];
          b1 += diff * Ixw[t]; b2 += diff * Iyw[t];
        }
      }
      b1 = wave_sum(b1); b2 = wave_sum(b2);
      const double dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
      nx += dx; ny += dy;
      if (dx * dx + dy * dy <= kEps * kEps) break;
      if (j > 0 && fabs(dx + pdx) < 0.01 && fabs(dy + pdy) < 0.01) { nx -= dx * 0.5; ny -= dy * 0.5; break; }
      pdx = dx; pdy = dy;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) { out[n] = make_float2((float)nx, (float)ny); status[n] = ok ? 1 : 0; }
}

// ordered compaction of the tracked pairs (n <= 1024), one workgroup
__global__ __launch_bounds__(1024) void compact_kernel(const float2* __restrict__ prev, const float2* __restrict__ next, const int* __restrict__ status,
                                                       const int* __restrict__ n_pts, float4* __restrict__ pairs, GmcResult* __restrict__ res) {
  __shared__ int s_scan[1024];
  const int tid = threadIdx.x, n = *n_pts;
  const int flag = (tid < n && status[tid]) ? 1 : 0;
  s_scan[tid] = flag;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  if (flag) pairs[s_scan[tid] - 1] = make_float4(prev[tid].x, prev[tid].y, next[tid].x, next[tid].y);
  if (tid == 1023) { res->n_prev = n; res->n_valid = s_scan[1023]; res->best_count = -1; res->a = 1; res->b = 0; res->tx = 0; res->ty = 0; }
}

__device__ __forceinline__ unsigned hash_u32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one wave per two-point similarity hypothesis: model + inlier count
__global__ __launch_bounds__(256) void ransac_kernel(const float4* __restrict__ pairs, const GmcResult* __restrict__ res, unsigned seed,
                                                     double4* __restrict__ model, int* __restrict__ count) {
  const int hyp = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = res->n_valid;
  int cnt = -1;
  double a = 1, b = 0, tx = 0, ty = 0;
  if (n >= 2) {
    const int i = (int)(hash_u32(seed ^ hash_u32(2u * hyp)) % (unsigned)n), j = (int)(hash_u32(seed ^ hash_u32(2u * hyp + 1u)) % (unsigned)n);
    const double pix = pairs[i].x, piy = pairs[i].y, qix = pairs[i].z, qiy = pairs[i].w;
    const double dpx = (double)pairs[j].x - pix, dpy = (double)pairs[j].y - piy;
    const double den = dpx * dpx + dpy * dpy;
    if (i != j && den >= 1e-12) {
      const double dqx = (double)pairs[j].z - qix, dqy = (double)pairs[j].w - qiy;
      a = (dpx * dqx + dpy * dqy) / den; b = (dpx * dqy - dpy * dqx) / den;
      tx = qix - (a * pix - b * piy); ty = qiy - (b * pix + a * piy);
      int c = 0;
      for (int k = lane; k < n; k += 64) {
        const double x = pairs[k].x, y = pairs[k].y;
        const double ex = a * x - b * y + tx - (double)pairs[k].z, ey = b * x + a * y + ty - (double)pairs[k].w;
        c += (ex * ex + ey * ey < kRansacThr * kRansacThr) ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
      cnt = c;
    }
  }
  if (lane == 0) { model[hyp] = make_double4(a, b, tx, ty); count[hyp] = cnt; }
}

__global__ __launch_bounds__(kHyp) void argmax_kernel(const double4* __restrict__ model, const int* __restrict__ count, GmcResult* __restrict__ res) {
  __shared__ int s_c[kHyp], s_i[kHyp];
  const int tid = threadIdx.x;
  s_c[tid] = count[tid]; s_i[tid] = tid;
  __syncthreads();
  for (int o = kHyp / 2; o >= 1; o >>= 1) {
    if (tid < o) {
      const int c2 = s_c[tid + o], i2 = s_i[tid + o];
      if (c2 > s_c[tid] || (c2 == s_c[tid] && i2 < s_i[tid])) { s_c[tid] = c2; s_i[tid] = i2; }
    }
    __syncthreads();
  }
  if (tid == 0 && s_c[0] >= 0) {
    const double4 m = model[s_i[0]];
    res->best_count = s_c[0]; res->a = m.x; res->b = m.y; res->tx = m.z; res->ty = m.w;
  }
}

bool similarity_from(const std::vector<float4>& pr, const std::vector<char>& inl, double M[6]) {
  double mpx = 0, mpy = 0, mqx = 0, mqy = 0;
  int n = 0;
  for (size_t i = 0; i < pr.size(); ++i)
    if (inl[i]) { mpx += pr[i].x; mpy += pr[i].y; mqx += pr[i].z; mqy += pr[i].w; ++n; }
  if (n < 2) return false;
  mpx /= n; mpy /= n; mqx /= n; mqy /= n;
  double den = 0, sa = 0, sb = 0;
  for (size_t i = 0; i < pr.size(); ++i)
    if (inl[i]) {
      const double px = pr[i].x - mpx, py = pr[i].y - mpy, qx = pr[i].z - mqx, qy = pr[i].w - mqy;
      den += px * px + py * py; sa += px * qx + py * qy; sb += px * qy - py * qx;
    }
  if (den <= 1e-12) return false;
  const double a = sa / den, b = sb / den;
  M[0] = a; M[1] = -b; M[2] = mqx - (a * mpx - b * mpy);
  M[3] = b; M[4] = a; M[5] = mqy - (b * mpx + a * mpy);
  return true;
}

}  // namespace

struct Gmc::Impl {
  int device;
  // Two streams: frame t works on st[t & 1]. Pyramid + corner detection of a frame need nothing from other
  // frames, so consecutive frames overlap; only the flow step waits (events) for the previous frame's corners,
  // and a frame's buffers are reused two frames later, after the flow step that read them as "previous".
  hipStream_t st[2] = {nullptr, nullptr};
  bool own_st1 = false;
  hipEvent_t front_ev[2] = {nullptr, nullptr};   // pyramid + corners of the last frame of that parity are ready
  hipEvent_t back_ev[2] = {nullptr, nullptr};    // flow step of the last frame of that parity is done
  int w, h;                    // half-resolution gray size
  unsigned seed;
  DevBuf frame, gray[2], pyr[2], lam[2], cand[2], counters[2], hist16[2], pts[2], npts[2], next[2], status[2], pairs[2], res[2], model[2],
      count[2];
  Pyr P[2]{};
  int cur = 0;                 // index of the buffers the next frame is written to
  bool have_prev = false;
  // results of up to kRing submitted frames wait in pinned memory, one event each (stream order
  // makes the device-side buffers reusable from one frame to the next)
  static constexpr int kRing = 64;
  GmcResult* h_res = nullptr;  // pinned [kRing]
  float4* h_pairs = nullptr;   // pinned [kRing][1024]
  hipEvent_t done[kRing] = {};
  bool first[kRing] = {};
  // result ring, single producer (submit) / single consumer (collect): the two may run on different host threads
  std::atomic<unsigned> submitted{0}, collected{0};
  int pending() const { return (int)(submitted.load(std::memory_order_acquire) - collected.load(std::memory_order_acquire)); }
  int cand_cap = 0;
  int stats[3] = {0, 0, 0};
};

Gmc::Gmc(int device, hipStream_t stream, int gray_h, int gray_w, int seed) : impl_(new Impl) {
  Impl& S = *impl_;
  GTX_CHECK(gray_h >= 64 && gray_w >= 64, "gmc: gray image %dx%d too small", gray_w, gray_h);
  S.device = device; S.st[0] = stream; S.w = gray_w; S.h = gray_h; S.seed = (unsigned)seed;
  GTX_HIP(hipSetDevice(device));
  GTX_HIP(hipStreamCreateWithFlags(&S.st[1], hipStreamNonBlocking));   // default priority: a high-priority pair measured 830 vs 1060 frames/s
  S.own_st1 = true;
  for (int k = 0; k < 2; ++k) {
    GTX_HIP(hipEventCreateWithFlags(&S.front_ev[k], hipEventDisableTiming));
    GTX_HIP(hipEventCreateWithFlags(&S.back_ev[k], hipEventDisableTiming));
  }
  size_t total = 0;
  int w = gray_w, h = gray_h;
  for (int l = 0; l <= kMaxLevel; ++l) { total += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
  for (int k = 0; k < 2; ++k) {
    S.pyr[k].alloc(total);
    uint8_t* p = S.pyr[k].as<uint8_t>();
    w = gray_w; h = gray_h;
    for (int l = 0; l <= kMaxLevel; ++l) { S.P[k].img[l] = p; S.P[k].w[l] = w; S.P[k].h[l] = h; p += (size_t)w * h; w = (w + 1) / 2; h = (h + 1) / 2; }
    S.pts[k].alloc(sizeof(float2) * 1024);
    S.npts[k].alloc(sizeof(int));
    GTX_HIP(hipMemset(S.npts[k].p, 0, sizeof(int)));
  }
  S.cand_cap = gray_w * gray_h / 4;
  for (int k = 0; k < 2; ++k) {
    S.lam[k].alloc(sizeof(double) * gray_w * gray_h);
    S.cand[k].alloc(sizeof(Cand) * S.cand_cap);
    S.counters[k].alloc(16);
    S.hist16[k].alloc(sizeof(int) * 256);
    GTX_HIP(hipMemset(S.counters[k].p, 0, 16));
    GTX_HIP(hipMemset(S.hist16[k].p, 0, sizeof(int) * 256));
    S.next[k].alloc(sizeof(float2) * 1024); S.status[k].alloc(sizeof(int) * 1024); S.pairs[k].alloc(sizeof(float4) * 1024);
    S.res[k].alloc(sizeof(GmcResult)); S.model[k].alloc(sizeof(double4) * kHyp); S.count[k].alloc(sizeof(int) * kHyp);
  }
  GTX_HIP(hipHostMalloc((void**)&S.h_res, sizeof(GmcResult) * Impl::kRing));
  GTX_HIP(hipHostMalloc((void**)&S.h_pairs, sizeof(float4) * 1024 * Impl::kRing));
  for (auto& e : S.done) GTX_HIP(hipEventCreateWithFlags(&e, wait_event_flags(false)));
  GTX_HIP(hipDeviceSynchronize());
}

Gmc::~Gmc() {
  if (impl_) {
    if (impl_->h_res) (void)hipHostFree(impl_->h_res);
    if (impl_->h_pairs) (void)hipHostFree(impl_->h_pairs);
    for (auto& e : impl_->done)
      if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
      if (impl_->front_ev[k]) (void)hipEventDestroy(impl_->front_ev[k]);
      if (impl_->back_ev[k]) (void)hipEventDestroy(impl_->back_ev[k]);
    }
    if (impl_->own_st1 && impl_->st[1]) { (void)hipStreamSynchronize(impl_->st[1]); (void)hipStreamDestroy(impl_->st[1]); }
  }
}

void Gmc::reset() {
  GTX_CHECK(impl_->pending() == 0, "gmc: reset while a frame is in flight");
  impl_->have_prev = false;
}

void Gmc::restart() { impl_->have_prev = false; }   // submit-side state only

void Gmc::submit_gray_dev(const void* gray, int gh, int gw) {
  Impl& S = *impl_;
  GTX_CHECK(gray && gh == S.h && gw == S.w, "gmc: gray image is %dx%d, expected %dx%d", gw, gh, S.w, S.h);
  GTX_CHECK(S.pending() < Impl::kRing, "gmc: %d frames already in flight, collect first", S.pending());
  const int slot = (int)(S.submitted.load(std::memory_order_relaxed) % Impl::kRing);
  GTX_HIP(hipSetDevice(S.device));
  const int c = S.cur, p = c ^ 1;
  hipStream_t s = S.st[c];
  const Pyr& Pc = S.P[c];
  // this parity's pyramid and corners were last read, as "previous", by the flow step of the frame before this one
  GTX_HIP(hipStreamWaitEvent(s, S.back_ev[p], 0));
  // pyramid of the current frame (level 0 is a copy: the caller's buffer may be recycled)
  GTX_HIP(hipMemcpyAsync(const_cast<uint8_t*>(Pc.img[0]), gray, (size_t)S.w * S.h, hipMemcpyDeviceToDevice, s));
  for (int l = 1; l <= kMaxLevel; ++l)
    hipLaunchKernelGGL(pyrdown_kernel, dim3(cdiv(Pc.w[l], 256), Pc.h[l]), dim3(256), 0, s, Pc.img[l - 1], Pc.w[l - 1], Pc.h[l - 1],
                       const_cast<uint8_t*>(Pc.img[l]), Pc.w[l], Pc.h[l]);
  // corners of the current frame
  unsigned long long* max_bits = S.counters[c].as<unsigned long long>();
  int* n_cand = reinterpret_cast<int*>(max_bits + 1);
  hipLaunchKernelGGL(response_kernel, dim3(cdiv(S.w, kRT_W), cdiv(S.h, kRT_H)), dim3(256), 0, s, Pc.img[0], S.w, S.h, S.lam[c].as<double>(), max_bits);
  hipLaunchKernelGGL(nms_kernel, dim3(cdiv(S.w - 2, 256), cdiv(S.h - 2, kNmsRows)), dim3(256), 0, s, S.lam[c].as<double>(), S.w, S.h, max_bits, S.cand[c].as<Cand>(),
                     n_cand, S.cand_cap, S.hist16[c].as<int>());
  hipLaunchKernelGGL(select_kernel, dim3(1), dim3(1024), 0, s, S.cand[c].as<Cand>(), n_cand, S.cand_cap, S.w, S.hist16[c].as<int>(), max_bits,
                     S.pts[c].as<float2>(), S.npts[c].as<int>());
  GTX_HIP(hipEventRecord(S.front_ev[c], s));
  S.first[slot] = !S.have_prev;
  if (S.have_prev) {
    GTX_HIP(hipStreamWaitEvent(s, S.front_ev[p], 0));          // the previous frame's pyramid and corners (other stream)
    hipLaunchKernelGGL(lk_kernel, dim3(cdiv(kMaxCorners, 4)), dim3(256), 0, s, S.P[p], Pc, S.pts[p].as<float2>(), S.npts[p].as<int>(),
                       S.next[c].as<float2>(), S.status[c].as<int>());
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, s, S.pts[p].as<float2>(), S.next[c].as<float2>(), S.status[c].as<int>(),
                       S.npts[p].as<int>(), S.pairs[c].as<float4>(), S.res[c].as<GmcResult>());
    hipLaunchKernelGGL(ransac_kernel, dim3(kHyp / 4), dim3(256), 0, s, S.pairs[c].as<float4>(), S.res[c].as<GmcResult>(), S.seed,
                       S.model[c].as<double4>(), S.count[c].as<int>());
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(kHyp), 0, s, S.model[c].as<double4>(), S.count[c].as<int>(), S.res[c].as<GmcResult>());
    GTX_HIP(hipMemcpyAsync(S.h_res + slot, S.res[c].p, sizeof(GmcResult), hipMemcpyDeviceToHost, s));
    GTX_HIP(hipMemcpyAsync(S.h_pairs + (size_t)slot * 1024, S.pairs[c].p, sizeof(float4) * 1024, hipMemcpyDeviceToHost, s));
  }
  GTX_HIP(hipEventRecord(S.back_ev[c], s));                    // whatever read the other parity's buffers has been queued
  GTX_HIP(hipGetLastError());
  GTX_HIP(hipEventRecord(S.done[slot], s));
  S.submitted.fetch_add(1, std::memory_order_release);
  S.have_prev = true;
  S.cur = p;       // the next frame overwrites what is now "previous"
}

void Gmc::submit_frame(const uint8_t* frame_bgr, int h, int w) {
  Impl& S = *impl_;
  GTX_CHECK(frame_bgr && h / 2 == S.h && w / 2 == S.w, "gmc: frame is %dx%d, created for %dx%d", w, h, 2 * S.w, 2 * S.h);
  GTX_CHECK(S.pending() == 0, "gmc: the host-frame path keeps one frame in flight");
  GTX_HIP(hipSetDevice(S.device));
  const size_t bytes = (size_t)h * w * 3;
  if (S.frame.bytes < bytes) S.frame.alloc(bytes);
  DevBuf& gray = S.gray[S.cur];
  hipStream_t s = S.st[S.cur];
  if (gray.bytes < (size_t)S.w * S.h) gray.alloc((size_t)S.w * S.h);
  GTX_HIP(hipMemcpyAsync(S.frame.p, frame_bgr, bytes, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(gray_half_kernel, dim3(cdiv(S.w, 256), S.h), dim3(256), 0, s, S.frame.as<uint8_t>(), w, gray.as<uint8_t>(), S.h, S.w);
  submit_gray_dev(gray.p, S.h, S.w);
}

void Gmc::submit_frame_dev(const void* frame_bgr_dptr, int h, int w, bool restart) {
  Impl& S = *impl_;
  GTX_CHECK(frame_bgr_dptr && h / 2 == S.h && w / 2 == S.w, "gmc: frame is %dx%d, created for %dx%d", w, h, 2 * S.w, 2 * S.h);
  GTX_HIP(hipSetDevice(S.device));
  // one gray buffer per parity (= per stream): conversion, the copy into the pyramid and the next conversion on
  // that stream are ordered. Allocated on first use, before anything is queued on it.
  DevBuf& gbuf = S.gray[S.cur];
  if (gbuf.bytes < (size_t)S.w * S.h) gbuf.alloc((size_t)S.w * S.h);
  uint8_t* gray = gbuf.as<uint8_t>();
  hipLaunchKernelGGL(gray_half_kernel, dim3(cdiv(S.w, 256), S.h), dim3(256), 0, S.st[S.cur], static_cast<const uint8_t*>(frame_bgr_dptr), w, gray, S.h, S.w);
  if (restart) S.have_prev = false;
  submit_gray_dev(gray, S.h, S.w);
}

void Gmc::collect(double A[6], int* valid, int stats[3]) {
  Impl& S = *impl_;
  GTX_CHECK(S.pending() > 0, "gmc: collect without a submitted frame");
  GTX_HIP(hipSetDevice(S.device));
  const int slot = (int)(S.collected.load(std::memory_order_relaxed) % Impl::kRing);
  GTX_HIP(hipEventSynchronize(S.done[slot]));
  const double I6[6] = {1, 0, 0, 0, 1, 0};
  std::memcpy(A, I6, sizeof I6);
  if (valid) *valid = 0;
  S.stats[0] = S.stats[1] = S.stats[2] = 0;
  if (!S.first[slot]) {
    const GmcResult& R = S.h_res[slot];
    S.stats[0] = R.n_prev; S.stats[1] = R.n_valid;
    if (R.n_valid > 4 && R.best_count >= 0) {
      const float4* hp = S.h_pairs + (size_t)slot * 1024;
      std::vector<float4> pr(hp, hp + R.n_valid);
      double M[6] = {R.a, -R.b, R.tx, R.b, R.a, R.ty};
      std::vector<char> inl(pr.size());
      int n_inl = 0;
      for (int it = 0; it < 3; ++it) {
        n_inl = 0;
        for (size_t i = 0; i < pr.size(); ++i) {
          const double ex = M[0] * pr[i].x + M[1] * pr[i].y + M[2] - pr[i].z, ey = M[3] * pr[i].x + M[4] * pr[i].y + M[5] - pr[i].w;
          inl[i] = (ex * ex + ey * ey < kRansacThr * kRansacThr) ? 1 : 0;
          n_inl += inl[i];
        }
        if (n_inl < 2) break;
        double M2[6];
        if (!similarity_from(pr, inl, M2)) break;
        std::memcpy(M, M2, sizeof M);
      }
      M[2] *= 2.0; M[5] *= 2.0;       // back to full-resolution pixels (downscale 2)
      std::memcpy(A, M, sizeof M);
      S.stats[2] = n_inl;
      if (valid) *valid = 1;
    }
  }
  if (stats) std::memcpy(stats, S.stats, sizeof S.stats);
  S.collected.fetch_add(1, std::memory_order_release);   // the slot's pinned records are free for the producer again
}

void Gmc::debug_points(int which, int cap, int* n, float* xy, int* status) const {
  const Impl& S = *impl_;
  GTX_CHECK(S.pending() == 0, "gmc: debug read while a frame is in flight");
  GTX_HIP(hipSetDevice(S.device));
  GTX_HIP(hipStreamSynchronize(S.st[0]));
  GTX_HIP(hipStreamSynchronize(S.st[1]));
  // which 0: corners of the last submitted frame; 1: corners of the frame before; 2: their LK positions in the last frame
  const int last = S.cur ^ 1, before = S.cur;
  const DevBuf& np = which == 0 ? S.npts[last] : S.npts[before];
  int cnt = 0;
  GTX_HIP(hipMemcpy(&cnt, np.p, sizeof(int), hipMemcpyDeviceToHost));
  *n = cnt;
  const int m = std::min(cnt, cap);
  if (m <= 0) return;
  const void* src = which == 0 ? S.pts[last].p : which == 1 ? S.pts[before].p : S.next[last].p;
  if (xy) GTX_HIP(hipMemcpy(xy, src, sizeof(float2) * m, hipMemcpyDeviceToHost));
  if (status && which == 2) GTX_HIP(hipMemcpy(status, S.status[last].p, sizeof(int) * m, hipMemcpyDeviceToHost));
}

}  // namespace gtx
