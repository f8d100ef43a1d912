// RT-DETR's non-convolution kernels (see rtdetr_kernels.hpp). gfx950 only.
//
// What bounds them at the reference configuration (one 3840x2160 frame stretched to 1920 x 1920, rtdetr-l):
//   rt_stem1 / rt_pool2 / rt_dwconv / rt_upsample2x / rt_tokens_in / rt_layernorm / rt_mask_invalid   HBM (each tensor read and written once)
//   rt_linear                     the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 157 TFLOP/s dense); tiny M: launch latency
//   rt_mha                        vector fp32 (2 d FMAs per query-key pair from LDS broadcasts)
//   rt_topk / rt_gather / rt_refer / rt_deform / rt_post   latency (a few hundred queries per image)
#include "rtdetr_kernels.hpp"

#include <cfloat>

namespace gtx {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- 8-channel groups in the three activation formats; e = element index of the group's first channel (a multiple of 8)
template <int FMT> __device__ __forceinline__ void load8(const void* base, size_t e, float v[8]) {
  const char* b = static_cast<const char*>(base);
  if constexpr (FMT == DT_F16) {
    const half8 h = *reinterpret_cast<const half8*>(b + e * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
  } else if constexpr (FMT == DT_F32) {
    const float4 a = *reinterpret_cast<const float4*>(b + e * 4), c = *reinterpret_cast<const float4*>(b + e * 4 + 16);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  } else {                                                       // pair format: 8 hi halves then 8 lo halves (split_format.hpp)
    const half8 hi = *reinterpret_cast<const half8*>(b + e * 4), lo = *reinterpret_cast<const half8*>(b + e * 4 + 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)hi[i] + (float)lo[i];
  }
}
template <int FMT> __device__ __forceinline__ void store8(void* base, size_t e, const float v[8], bool& sat) {
  char* b = static_cast<char*>(base);
  if constexpr (FMT == DT_F16) {
    half8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (_Float16)v[i];
    *reinterpret_cast<half8*>(b + e * 2) = h;
  } else if constexpr (FMT == DT_F32) {
    *reinterpret_cast<float4*>(b + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(b + e * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    half8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = __builtin_amdgcn_fmed3f(v[i], -65504.f, 65504.f);   // a NaN comes out as -65504, like the convolutions' split
      sat |= x != v[i];
      hi[i] = (_Float16)x;
      lo[i] = (_Float16)(x - (float)hi[i]);                                // x - hi is exact in fp32
    }
    *reinterpret_cast<half8*>(b + e * 4) = hi;
    *reinterpret_cast<half8*>(b + e * 4 + 16) = lo;
  }
}
template <int FMT> __device__ __forceinline__ float load1(const void* base, size_t e) {
  const char* b = static_cast<const char*>(base);
  if constexpr (FMT == DT_F16) return (float)*reinterpret_cast<const _Float16*>(b + e * 2);
  else if constexpr (FMT == DT_F32) return *reinterpret_cast<const float*>(b + e * 4);
  else {
    const char* g = b + (e & ~(size_t)7) * 4 + 2 * (e & 7);
    return (float)*reinterpret_cast<const _Float16*>(g) + (float)*reinterpret_cast<const _Float16*>(g + 16);
  }
}
constexpr __host__ __device__ int fmt_size(int fmt) { return fmt == DT_F16 ? 2 : 4; }

#define RT_FMT(fmt, ...)                                                           \
  do {                                                                             \
    switch (fmt) {                                                                 \
      case DT_F16: { constexpr int F = DT_F16; __VA_ARGS__; } break;               \
      case DT_F32: { constexpr int F = DT_F32; __VA_ARGS__; } break;               \
      case DT_F32S: { constexpr int F = DT_F32S; __VA_ARGS__; } break;             \
      default: fail(-3, "rtdetr: unknown activation format %d", (int)(fmt));       \
    }                                                                              \
    GTX_HIP(hipGetLastError());                                                    \
  } while (0)

__device__ __forceinline__ void flag_sat(int* sat, bool s) {
  if (sat && s) *sat = 1;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ============================================================================ HGStem.stem1
template <int FMT>
__global__ __launch_bounds__(256) void rt_stem1_kernel(const uchar4* __restrict__ img, int N, int H, int W, const float* __restrict__ w27,
                                                       const float* __restrict__ bias, RtMap out, int* sat) {
  // ultralytics' `im.float() / 255` has 256 possible results: one correctly rounded division per table entry instead of 27 x 3 per thread
  __shared__ float s_lut[256];
  {
    float f = (float)threadIdx.x / 255.f;
    if constexpr (FMT == DT_F16) f = (float)(_Float16)f;
    s_lut[threadIdx.x] = f;
  }
  __syncthreads();
  // One thread per output pixel, every channel: the weight addresses are then the same for the whole wave (scalar loads, the
  // multiplier straight from a scalar register). With a channel group per thread they were 81 x 8 vector loads per thread and the
  // kernel ran at 1 TB/s of stores; the nine image words and their table look-ups are now made once per pixel, too.
  const int c0 = out.c;
  const size_t total = (size_t)N * out.h * out.w;
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= total) return;
  const int ox = (int)(pix % out.w), oy = (int)((pix / out.w) % out.h), n = (int)(pix / ((size_t)out.w * out.h));
  const uchar4* base = img + (size_t)n * H * W;
  float px[27];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const uchar4 u = in ? base[(size_t)iy * W + ix] : make_uchar4(0, 0, 0, 0);
      px[(ky * 3 + kx) * 3 + 0] = in ? s_lut[u.x] : 0.f;      // zero padding is zero AFTER the division (0 / 255 = 0 anyway)
      px[(ky * 3 + kx) * 3 + 1] = in ? s_lut[u.y] : 0.f;
      px[(ky * 3 + kx) * 3 + 2] = in ? s_lut[u.z] : 0.f;
    }
  bool s = false;
  for (int h16 = 0; h16 < c0 / 16; ++h16) {
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[h16 * 16 + j];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float* wt = w27 + (size_t)k * c0 + h16 * 16;       // wave-uniform
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = fmaf(px[k], wt[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = fmaxf(acc[j], 0.f);
    store8<FMT>(out.ptr, pix * out.cstride + out.coff + h16 * 16, acc, s);
    store8<FMT>(out.ptr, pix * out.cstride + out.coff + h16 * 16 + 8, acc + 8, s);
  }
  flag_sat(sat, s);
}

// ============================================================================ 2x2 stride-1 max pool on the zero-padded map
template <int FMT>
__global__ __launch_bounds__(256) void rt_pool2_kernel(RtMap in, RtMap out, int N, int* sat) {
  const int groups = in.c / 8;
  const size_t total = (size_t)N * in.h * in.w * groups;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % groups);
  const size_t pix = t / groups;
  const int x = (int)(pix % in.w), y = (int)((pix / in.w) % in.h);
  float m[8], v[8];
  load8<FMT>(in.ptr, pix * in.cstride + in.coff + g * 8, m);
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const int dy = k >> 1, dx = k & 1;
    if (y + dy < in.h && x + dx < in.w) {
      load8<FMT>(in.ptr, (pix + (size_t)dy * in.w + dx) * in.cstride + in.coff + g * 8, v);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;                   // F.pad's zeros take part in the maximum
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
  }
  bool s = false;
  store8<FMT>(out.ptr, pix * out.cstride + out.coff + g * 8, m, s);
  flag_sat(sat, s);
}

// ============================================================================ depthwise convolution
template <int FMT, int K>
__global__ __launch_bounds__(256) void rt_dwconv_kernel(RtMap in, RtMap out, int N, int stride, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int act, int* sat) {
  const int groups = in.c / 8, C = in.c;
  const size_t total = (size_t)N * out.h * out.w * groups;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % groups);
  const size_t pix = t / groups;
  const int ox = (int)(pix % out.w), oy = (int)((pix / out.w) % out.h), n = (int)(pix / ((size_t)out.w * out.h));
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[g * 8 + j] : 0.f;
  // branch-free taps: out-of-range ones read a clamped address and are multiplied by zero weights, so that a row's K loads
  // go out together instead of one behind each bounds test (the kernel is latency-bound: 25 small loads per thread)
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * stride - K / 2 + ky;
    const bool oky = iy >= 0 && iy < in.h;
    const int iyc = min(max(iy, 0), in.h - 1);
    float v[K][8];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ixc = min(max(ox * stride - K / 2 + kx, 0), in.w - 1);
      load8<FMT>(in.ptr, (((size_t)n * in.h + iyc) * in.w + ixc) * in.cstride + in.coff + g * 8, v[kx]);
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * stride - K / 2 + kx;
      const float m = (oky && ix >= 0 && ix < in.w) ? 1.f : 0.f;
      const float4 w0 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + g * 8);
      const float4 w1 = *reinterpret_cast<const float4*>(w + (size_t)(ky * K + kx) * C + g * 8 + 4);
      acc[0] = fmaf(v[kx][0], w0.x * m, acc[0]); acc[1] = fmaf(v[kx][1], w0.y * m, acc[1]); acc[2] = fmaf(v[kx][2], w0.z * m, acc[2]); acc[3] = fmaf(v[kx][3], w0.w * m, acc[3]);
      acc[4] = fmaf(v[kx][4], w1.x * m, acc[4]); acc[5] = fmaf(v[kx][5], w1.y * m, acc[5]); acc[6] = fmaf(v[kx][6], w1.z * m, acc[6]); acc[7] = fmaf(v[kx][7], w1.w * m, acc[7]);
    }
  }
  if (act == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
  }
  bool s = false;
  store8<FMT>(out.ptr, pix * out.cstride + out.coff + g * 8, acc, s);
  flag_sat(sat, s);
}

// Stride-1 form for channel counts that are multiples of 32 (HGBlock's LightConv 5x5 layers: 24 launches per pass): a workgroup
// takes an 8 x 8 pixel tile x 32 channels, stages the (8 + K - 1)^2 input patch ONCE in LDS as fp32 (the per-thread form read and
// converted every input pixel K^2 times: 25 small loads per thread made it latency-bound at 0.7 TB/s) and its K^2 x 32 weights,
// then every thread runs the same K^2 fused multiply-adds in the same order: the results are the per-thread kernel's bit for bit.
// (A 16 x 16-pixel tile with four pixels of a row per thread -- 26 LDS reads per 160 multiply-adds instead of 80, 1.56 x instead of 2.25 x
// halo -- was measured at the end of round 6: 0.609 against 0.59-0.60 ms for the 24 launches. Not LDS reads; the staging's round trip per workgroup.)
template <int FMT, int K>
__global__ __launch_bounds__(256) void rt_dwconv_tile_kernel(RtMap in, RtMap out, const float* __restrict__ w, const float* __restrict__ bias, int act, int* sat) {
  constexpr int T = 8, P = T + K - 1, R = K / 2;
  __shared__ float s_in[P * P][32];
  __shared__ float s_w[K * K][32];
  const int tiles_x = (out.w + T - 1) / T;
  const int tx0 = (blockIdx.x % tiles_x) * T, ty0 = (blockIdx.x / tiles_x) * T;
  const int c0 = blockIdx.y * 32, n = blockIdx.z, C = in.c;
  for (int i = threadIdx.x; i < P * P * 4; i += 256) {
    const int g = i & 3, pp = i >> 2;
    const int y = ty0 - R + pp / P, x = tx0 - R + pp % P;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < in.h && x >= 0 && x < in.w) load8<FMT>(in.ptr, (((size_t)n * in.h + y) * in.w + x) * in.cstride + in.coff + c0 + g * 8, v);
    *reinterpret_cast<float4*>(&s_in[pp][g * 8]) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(&s_in[pp][g * 8 + 4]) = make_float4(v[4], v[5], v[6], v[7]);
  }
  for (int i = threadIdx.x; i < K * K * 32; i += 256) s_w[i >> 5][i & 31] = w[(size_t)(i >> 5) * C + c0 + (i & 31)];
  __syncthreads();
  const int g = threadIdx.x & 3, px = (threadIdx.x >> 2) & 7, py = threadIdx.x >> 5;
  const int ox = tx0 + px, oy = ty0 + py;
  if (ox >= out.w || oy >= out.h) return;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[c0 + g * 8 + j] : 0.f;
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float* v = &s_in[(py + ky) * P + px + kx][g * 8];
      const float* ww = &s_w[ky * K + kx][g * 8];
      const float4 v0 = *reinterpret_cast<const float4*>(v), v1 = *reinterpret_cast<const float4*>(v + 4);
      const float4 w0 = *reinterpret_cast<const float4*>(ww), w1 = *reinterpret_cast<const float4*>(ww + 4);
      acc[0] = fmaf(v0.x, w0.x, acc[0]); acc[1] = fmaf(v0.y, w0.y, acc[1]); acc[2] = fmaf(v0.z, w0.z, acc[2]); acc[3] = fmaf(v0.w, w0.w, acc[3]);
      acc[4] = fmaf(v1.x, w1.x, acc[4]); acc[5] = fmaf(v1.y, w1.y, acc[5]); acc[6] = fmaf(v1.z, w1.z, acc[6]); acc[7] = fmaf(v1.w, w1.w, acc[7]);
    }
  if (act == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
  }
  bool s = false;
  store8<FMT>(out.ptr, (((size_t)n * out.h + oy) * out.w + ox) * out.cstride + out.coff + c0 + g * 8, acc, s);
  flag_sat(sat, s);
}

// ============================================================================ nearest 2x upsampling (raw 8-channel groups)
__global__ __launch_bounds__(256) void rt_upsample2x_kernel(RtMap in, RtMap out, int N, int gbytes) {
  const int groups = in.c / 8;
  const size_t total = (size_t)N * out.h * out.w * groups;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % groups);
  const size_t pix = t / groups;
  const int ox = (int)(pix % out.w), oy = (int)((pix / out.w) % out.h), n = (int)(pix / ((size_t)out.w * out.h));
  const size_t src = (((size_t)n * in.h + oy / 2) * in.w + ox / 2) * in.cstride + in.coff + g * 8;
  const size_t dst = pix * out.cstride + out.coff + g * 8;
  const int es = gbytes / 8;
  const uint4* sp = reinterpret_cast<const uint4*>(static_cast<const char*>(in.ptr) + src * es);
  uint4* dp = reinterpret_cast<uint4*>(static_cast<char*>(out.ptr) + dst * es);
  dp[0] = sp[0];
  if (gbytes == 32) dp[1] = sp[1];
}

// ============================================================================ map -> token rows (+ positional embedding)
template <int FMT>
__global__ __launch_bounds__(256) void rt_tokens_in_kernel(RtMap in, int N, const float* __restrict__ pos, float* __restrict__ src, float* __restrict__ q) {
  const int groups = in.c / 8, C = in.c;
  const size_t T = (size_t)in.h * in.w, total = (size_t)N * T * groups;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % groups);
  const size_t row = t / groups;
  float v[8];
  load8<FMT>(in.ptr, row * in.cstride + in.coff + g * 8, v);
  bool s = false;
  store8<DT_F32>(src, row * C + g * 8, v, s);
  if (q) {
    float p[8];
    load8<DT_F32>(pos, (row % T) * C + g * 8, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += p[j];
    store8<DT_F32>(q, row * C + g * 8, v, s);
  }
}

// ============================================================================ invalid anchors' rows -> 0
__device__ __forceinline__ bool anchor_valid(int level, int y, int x, int h, int w) {
  // RTDETRDecoder._generate_anchors: ((grid + 0.5) / [w, h], 0.05 * 2^level) all inside (eps, 1 - eps), eps = 0.01, in fp32
  const float cx = ((float)x + 0.5f) / (float)w, cy = ((float)y + 0.5f) / (float)h;
  const float wh = 0.05f * (float)(1 << level);
  const float eps = 1e-2f, hi = 0.99f;           // torch compares the fp32 tensor with the python doubles 0.01 and 0.99 (= 1 - eps) cast to fp32
  return cx > eps && cx < hi && cy > eps && cy < hi && wh > eps && wh < hi;
}
__global__ __launch_bounds__(256) void rt_mask_invalid_kernel(RtMap m, int N, int level, int gbytes) {
  const int groups = m.c / 8;
  const size_t total = (size_t)N * m.h * m.w * groups;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t % groups);
  const size_t pix = t / groups;
  const int x = (int)(pix % m.w), y = (int)((pix / m.w) % m.h);
  if (anchor_valid(level, y, x, m.h, m.w)) return;
  const int es = gbytes / 8;
  uint4* dp = reinterpret_cast<uint4*>(static_cast<char*>(m.ptr) + (pix * m.cstride + m.coff + g * 8) * es);
  dp[0] = make_uint4(0, 0, 0, 0);
  if (gbytes == 32) dp[1] = make_uint4(0, 0, 0, 0);
}

// ============================================================================ token linear layer on the fp32 matrix pipe
// A workgroup = 4 waves = 16 rows x 128 columns; a wave = 16 rows x 32 columns (two 16 x 16 accumulators). Per 16 values of K
// a lane holds X[row = l & 15][k0 + 4 (l >> 4) .. + 3] and, per column block, W[col = l & 15][the same four k]: MFMA j of the
// four sums the k set {4 kk + j}, the same in both operands (the order inside a dot product is free). W is read in its own
// row-major layout (16 bytes per lane, 64 contiguous bytes per column); the X tile goes through LDS once per workgroup.
constexpr int kLinKC = 128;                       // K values per step of a wave's weight fragments
constexpr int kLinSpan = 512;                     // K values of the X tile staged at a time (one staging round per 512: K <= 512 has one)
constexpr int kLinPitch = kLinSpan + 4;           // floats per staged row: rows 4 banks apart
// A workgroup = 16 rows x 64 columns, one 16-column block per wave (600 query rows are 38 row blocks: 64-column workgroups put 152
// of them on the chip for a 256-wide layer where 128-column ones put 76). The X tile of a span is staged once; inside it a wave
// streams its weight fragments one 128-value step ahead of the MFMAs that use them, so a step costs its 32 MFMAs (two
// accumulators: even and odd k groups, summed at the end) and not an L2 round trip -- the FFN's K = 1024 layer was 8 exposed
// round trips long.
__global__ __launch_bounds__(256) void rt_linear_kernel(const RtLinear p) {
  __shared__ float xs[16 * kLinPitch];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row0 = blockIdx.x * 16, col0 = blockIdx.y * 64 + wave * 16;
  const int lr = lane & 15, kk = lane >> 4;
  const bool cb = col0 < p.Nout;                                  // Nout % 16 == 0: whole column blocks
  const float* x2 = (p.x2 && (int)blockIdx.y * 64 < p.x2_cols) ? p.x2 : nullptr;   // the second addend feeds the first x2_cols columns only (x2_cols % 64 == 0)
  floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* w0 = p.w + (size_t)(col0 + lr) * p.K + 4 * kk;
  float4 bcur[kLinKC / 16], bnext[kLinKC / 16];
#define RT_LIN_LOAD_W(DST, KABS)                                                             \
  {                                                                                          \
    const int kc__ = min(kLinKC, p.K - (KABS));                                              \
    _Pragma("unroll") for (int i = 0; i < kLinKC / 16; ++i) {                                \
      DST[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                              \
      if (cb && 16 * i < kc__) DST[i] = *reinterpret_cast<const float4*>(w0 + (KABS) + 16 * i); \
    }                                                                                        \
  }
  for (int kb = 0; kb < p.K; kb += kLinSpan) {
    const int ks = min(kLinSpan, p.K - kb);                       // a multiple of 16
    RT_LIN_LOAD_W(bcur, kb)                                       // the span's first fragments: their L2 round trip runs under the staging
    __syncthreads();                                              // the previous span's reads are done
    for (int i = threadIdx.x; i < 16 * (kLinSpan / 4); i += 256) {
      const int r = i / (kLinSpan / 4), c4 = (i % (kLinSpan / 4)) * 4;
      if (c4 >= ks) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < p.M) {
        v = *reinterpret_cast<const float4*>(p.x + (size_t)(row0 + r) * p.ldx + kb + c4);
        if (x2) {
          const float4 u = *reinterpret_cast<const float4*>(x2 + (size_t)(row0 + r) * p.ldx2 + kb + c4);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
      }
      *reinterpret_cast<float4*>(&xs[r * kLinPitch + c4]) = v;
    }
    __syncthreads();
    for (int k0 = 0; k0 < ks; k0 += kLinKC) {
      const int kc = min(kLinKC, ks - k0);
      if (k0 + kLinKC < ks) RT_LIN_LOAD_W(bnext, kb + k0 + kLinKC)
      if (cb) {
#pragma unroll
        for (int i = 0; i < kLinKC / 16; ++i) {
          if (16 * i >= kc) break;
          const float4 a = *reinterpret_cast<const float4*>(&xs[lr * kLinPitch + k0 + 16 * i + 4 * kk]);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bcur[i].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bcur[i].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bcur[i].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bcur[i].w, acc1, 0, 0, 0);
        }
      }
      if (k0 + kLinKC < ks) {
#pragma unroll
        for (int i = 0; i < kLinKC / 16; ++i) bcur[i] = bnext[i];
      }
    }
  }
#undef RT_LIN_LOAD_W
  // C: lane l holds rows 4 (l >> 4) + i of column l & 15
  if (!cb) return;
  const int col = col0 + lr;
  const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 4 * kk + i;
    if (row >= p.M) continue;
    float v = (acc0[i] + acc1[i]) + bv;
    if (p.act == 2) v = fmaxf(v, 0.f);
    else if (p.act == 3) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));   // nn.GELU(): exact erf form
    if (p.res) v += p.res[(size_t)row * p.ldr + col];
    p.y[(size_t)row * p.ldy + col] = v;
  }
}

// ============================================================================ LayerNorm (one wave per row)
template <int FIN, int FOUT>
__global__ __launch_bounds__(256) void rt_layernorm_kernel(RtRows in, RtRows out, long rows, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int* sat) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int groups = C / 8;                       // <= 128: a lane holds at most two groups
  float v[2][8];
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int g = lane + 64 * r;
    if (g < groups) {
      load8<FIN>(in.ptr, (size_t)row * in.cstride + in.coff + g * 8, v[r]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[r][j];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r)
    if (lane + 64 * r < groups) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[r][j] - mean; sq = fmaf(d, d, sq); }
    }
  const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)C + 1e-5f);
  bool s = false;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int g = lane + 64 * r;
    if (g < groups) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[r][j] - mean) * rstd * gamma[g * 8 + j] + beta[g * 8 + j];
      store8<FOUT>(out.ptr, (size_t)row * out.cstride + out.coff + g * 8, o, s);
    }
  }
  flag_sat(sat, s);
}

// ============================================================================ multi-head attention on token rows
// A workgroup = 64 queries of one (image, head); lane = one query row (its q vector, running maximum, denominator and output
// row in registers); the four waves share the keys four ways -- each stages its quarter in tiles of 32 keys in its own LDS
// region and reads them as broadcasts -- and their partial softmaxes are merged through LDS at the end.
template <int D>
__global__ __launch_bounds__(256) void rt_mha_kernel(const float* __restrict__ qkv, int ld, int T, int C, float* __restrict__ out, int ldo) {
  constexpr int KT = 32;
  __shared__ float kv[4][2][KT][D];               // per wave: K tile, V tile
  __shared__ float part[3][64][D + 2];            // waves 1..3: (m, l, acc[D]) per query
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int head = blockIdx.y, n = blockIdx.z;
  const int qi = blockIdx.x * 64 + lane;
  const float* base = qkv + (size_t)n * T * ld;
  const float scale = 1.f / sqrtf((float)D);
  float q[D], acc[D];
  const bool live = qi < T;
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = *reinterpret_cast<const float4*>(base + (size_t)qi * ld + head * D + d);
    q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;   // torch scales q before the product
    acc[d] = acc[d + 1] = acc[d + 2] = acc[d + 3] = 0.f;
  }
  float m = -FLT_MAX, l = 0.f;
  const int per = (T + 3) / 4, k_begin = wave * per, k_end = min(T, k_begin + per);
  for (int k0 = k_begin; k0 < k_end; k0 += KT) {
    const int nk = min(KT, k_end - k0);
    // stage: KT keys x D floats of K and of V (each lane a float4 per step); wave-private region, no workgroup barrier
    for (int i = lane; i < KT * (D / 4); i += 64) {
      const int kr = i / (D / 4), d4 = (i % (D / 4)) * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (kr < nk) {
        const float* row = base + (size_t)(k0 + kr) * ld + head * D + d4;
        a = *reinterpret_cast<const float4*>(row + C);
        b = *reinterpret_cast<const float4*>(row + 2 * C);
      }
      *reinterpret_cast<float4*>(&kv[wave][0][kr][d4]) = a;
      *reinterpret_cast<float4*>(&kv[wave][1][kr][d4]) = b;
    }
    __builtin_amdgcn_s_waitcnt(0);                // the wave's own LDS writes before its reads
    __builtin_amdgcn_wave_barrier();
    for (int c0 = 0; c0 < nk; c0 += 8) {
      float s[8];
      float cm = m;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          const float4 kq = *reinterpret_cast<const float4*>(&kv[wave][0][c0 + j][d]);
          t = fmaf(q[d], kq.x, t); t = fmaf(q[d + 1], kq.y, t); t = fmaf(q[d + 2], kq.z, t); t = fmaf(q[d + 3], kq.w, t);
        }
        s[j] = (c0 + j < nk) ? t : -FLT_MAX;
        cm = fmaxf(cm, s[j]);
      }
      const float r = __expf(m - cm);
      l *= r;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] *= r;
      m = cm;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pj = (c0 + j < nk) ? __expf(s[j] - m) : 0.f;
        l += pj;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
          const float4 vq = *reinterpret_cast<const float4*>(&kv[wave][1][c0 + j][d]);
          acc[d] = fmaf(pj, vq.x, acc[d]); acc[d + 1] = fmaf(pj, vq.y, acc[d + 1]); acc[d + 2] = fmaf(pj, vq.z, acc[d + 2]); acc[d + 3] = fmaf(pj, vq.w, acc[d + 3]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (wave > 0) {
    part[wave - 1][lane][0] = m;
    part[wave - 1][lane][1] = l;
#pragma unroll
    for (int d = 0; d < D; ++d) part[wave - 1][lane][2 + d] = acc[d];
  }
  __syncthreads();
  if (wave == 0 && live) {
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      const float m2 = part[w][lane][0], l2 = part[w][lane][1];
      if (l2 == 0.f) continue;                    // that wave had no keys
      const float mn = fmaxf(m, m2), r1 = __expf(m - mn), r2 = __expf(m2 - mn);
      l = l * r1 + l2 * r2;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = acc[d] * r1 + part[w][lane][2 + d] * r2;
      m = mn;
    }
    const float inv = 1.f / l;
    float* o = out + ((size_t)n * T + qi) * ldo + head * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(o + d) = make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
  }
}

// Head dimension 32 (RT-DETR's 256 / 8) on the fp32 matrix pipe. A wave owns 16 queries and walks the keys 16 at a time:
//   S^T[key][query]  = sum_dim K[key][dim] Q[query][dim]         8 x v_mfma_f32_16x16x4_f32 (A = K rows from LDS, B = the wave's Q fragment)
//   O^T[dim][query] += sum_key V[key][dim] P^T[key][query]       8 x (two 16-dim tiles x four key groups)
// S^T comes out with the query on the lane's column and four keys (4 kk + r) in its registers -- exactly the B operand of the second
// product when MFMA r sums the key set {4 kk + r}: the probabilities never move between lanes (the order inside a sum is free), only the
// running maximum and denominator of a query are reduced over its four lanes (two shuffles each). The four waves of a workgroup share
// the K / V tiles (64 keys per stage) through LDS.
__global__ __launch_bounds__(256) void rt_mha32_kernel(const float* __restrict__ qkv, int ld, int T, int C, float* __restrict__ out, int ldo) {
  constexpr int D = 32, KS = 64, PITCH = 36;
  __shared__ float s_k[KS * PITCH];
  __shared__ float s_v[KS * PITCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lc = lane & 15, kk = lane >> 4;
  const int head = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const float* base = qkv + (size_t)n * T * ld + head * D;
  const float scale = 1.f / sqrtf((float)D);
  float qf[8];                                     // B operand of the first product: Q[query lc][4 i + kk], scaled
  {
    const int qi = q0 + lc;
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = qi < T ? base[(size_t)qi * ld + 4 * i + kk] * scale : 0.f;
  }
  floatx4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};   // O^T[dim 4 kk + r (+16)][query lc]
  float m = -FLT_MAX, l = 0.f;
  // stage s + 1's K / V rows are loaded into registers while stage s is multiplied (two float4 of each per thread): without it
  // every stage began with an exposed global round trip (57 of them for AIFI's 3 600 keys)
  constexpr int NLD = KS * (D / 4) / 256;          // 2
  float4 pk[NLD], pv[NLD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = threadIdx.x + 256 * j, kr = i / (D / 4), d4 = (i % (D / 4)) * 4;
      pk[j] = pv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + kr < T) {
        const float* row = base + (size_t)(k0 + kr) * ld + d4;
        pk[j] = *reinterpret_cast<const float4*>(row + C);
        pv[j] = *reinterpret_cast<const float4*>(row + 2 * C);
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < T; k0 += KS) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = threadIdx.x + 256 * j, kr = i / (D / 4), d4 = (i % (D / 4)) * 4;
      *reinterpret_cast<float4*>(&s_k[kr * PITCH + d4]) = pk[j];
      *reinterpret_cast<float4*>(&s_v[kr * PITCH + d4]) = pv[j];
    }
    if (k0 + KS < T) fetch(k0 + KS);
    __syncthreads();
    const int nk = min(KS, T - k0);
    for (int t0 = 0; t0 < nk; t0 += 16) {
      floatx4 sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(s_k[(t0 + lc) * PITCH + 4 * i + kk], qf[i], sc, 0, 0, 0);
      float mx = -FLT_MAX;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (t0 + 4 * kk + r >= nk) sc[r] = -FLT_MAX;          // keys past the end
        mx = fmaxf(mx, sc[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mn = fmaxf(m, mx), resc = __expf(m - mn);
      float p[4], ps = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = sc[r] > -FLT_MAX ? __expf(sc[r] - mn) : 0.f; ps += p[r]; }
      ps += __shfl_xor(ps, 16);
      ps += __shfl_xor(ps, 32);
      l = l * resc + ps;
      m = mn;
#pragma unroll
      for (int r = 0; r < 4; ++r) { o0[r] *= resc; o1[r] *= resc; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = &s_v[(t0 + 4 * kk + r) * PITCH];
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[lc], p[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[16 + lc], p[r], o1, 0, 0, 0);
      }
    }
  }
  const int qi = q0 + lc;
  if (qi < T) {
    const float inv = 1.f / l;
    float* o = out + ((size_t)n * T + qi) * ldo + head * D + 4 * kk;
    *reinterpret_cast<float4*>(o) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
    *reinterpret_cast<float4*>(o + 16) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
  }
}

// ============================================================================ query selection (top-k anchors per image)
__device__ __forceinline__ unsigned sortable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // larger float <=> larger unsigned (NaN aside)
}
__device__ __forceinline__ void level_of(const RtLevels& L, int a, int& lv, int& y, int& x) {
  lv = 0;
  while (lv + 1 < L.n_levels && a >= L.h[lv] * L.w[lv]) { a -= L.h[lv] * L.w[lv]; ++lv; }
  y = a / L.w[lv];
  x = a % L.w[lv];
}

// One workgroup of 1024 threads per image: keys = max class score per anchor; an 8-bit MSB-first radix select finds the nq-th
// largest key; the strictly larger ones and the first (lowest anchor index) of the equal ones are collected; a bitonic sort
// orders them (key descending, index ascending).
// the keys first, one anchor per thread over the whole chip (75 600 anchors per image at 1920^2: one workgroup walking them, 74 dependent
// round trips per thread, was half of the selection's time: 0.27 -> 0.13 ms)
template <int FMT>
__global__ __launch_bounds__(256) void rt_topk_keys_kernel(RtLevels sc, int nc, int S, unsigned* __restrict__ keys_all) {
  const int a = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (a >= S) return;
  int lv, y, x;
  level_of(sc, a, lv, y, x);
  const size_t e = (((size_t)n * sc.h[lv] + y) * sc.w[lv] + x) * sc.cstride[lv] + sc.coff[lv];
  float mx = load1<FMT>(sc.ptr[lv], e);
  for (int c = 1; c < nc; ++c) mx = fmaxf(mx, load1<FMT>(sc.ptr[lv], e + c));
  keys_all[(size_t)n * S + a] = sortable(mx);
}

__global__ __launch_bounds__(1024) void rt_topk_kernel(int S, int nq, const unsigned* __restrict__ keys_all, int* __restrict__ out_idx) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long items[1024];
  __shared__ unsigned s_prefix, s_need, s_count, s_eq_base;
  __shared__ unsigned wave_cnt[16];
  const int n = blockIdx.x, tid = threadIdx.x;
  const unsigned* keys = keys_all + (size_t)n * S;
  if (tid == 0) { s_prefix = 0; s_need = (unsigned)nq; }
  __syncthreads();
  unsigned mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    // a thread's consecutive candidates mostly share the bin (scores share their leading bytes): it counts a run privately and
    // adds once per run -- about one LDS atomic per thread and pass instead of 74 on one address
    unsigned cur = 0xFFFFFFFFu, cnt = 0;
    for (int a = tid; a < S; a += 1024) {
      const unsigned k = keys[a];
      if ((k & mask) != prefix) continue;
      const unsigned bin = (k >> shift) & 255u;
      if (bin == cur) { ++cnt; continue; }
      if (cnt) atomicAdd(&hist[cur], cnt);
      cur = bin;
      cnt = 1;
    }
    if (cnt) atomicAdd(&hist[cur], cnt);
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, b = 255;
      for (;; --b) {
        if (hist[b] >= need || b == 0) break;
        need -= hist[b];
      }
      s_need = need;                               // how many of bin b (and below within it) are still wanted
      s_prefix = prefix | (b << shift);
    }
    mask |= 255u << shift;
    __syncthreads();
  }
  const unsigned thr = s_prefix, need_eq = s_need;   // nq-th largest key; that many of the keys equal to it are taken
  if (tid == 0) { s_count = 0; s_eq_base = 0; }
  __syncthreads();
  // strictly larger: any order (sorted afterwards)
  for (int a = tid; a < S; a += 1024) {
    const unsigned k = keys[a];
    if (k > thr) {
      const unsigned slot = atomicAdd(&s_count, 1u);
      items[slot] = ((unsigned long long)k << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a);
    }
  }
  __syncthreads();
  const unsigned n_gt = s_count;
  // equal: in anchor order, the first need_eq of them
  for (int a0 = 0; a0 < S && s_eq_base < need_eq; a0 += 1024) {
    const int a = a0 + tid;
    const bool eq = a < S && keys[a] == thr;
    const unsigned long long bal = __ballot(eq);
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = s_eq_base;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
    if (eq && rank < need_eq) items[n_gt + rank] = ((unsigned long long)thr << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)a);
    __syncthreads();
    if (tid == 0) {
      unsigned tot = 0;
      for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
      s_eq_base += tot;
    }
    __syncthreads();
  }
  const unsigned have = n_gt + need_eq;            // == nq when S >= nq
  for (int i = tid; i < 1024; i += 1024)
    if ((unsigned)i >= have) items[i] = 0ull;
  __syncthreads();
  // bitonic sort, descending, 1024 slots
  for (int k = 2; k <= 1024; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = tid ^ j;
      if (ixj > tid) {
        const unsigned long long a = items[tid], b = items[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? a < b : a > b) { items[tid] = b; items[ixj] = a; }
      }
      __syncthreads();
    }
  if (tid < nq) out_idx[(size_t)n * nq + tid] = (int)(0xFFFFFFFFu - (unsigned)(items[tid] & 0xFFFFFFFFull));
}

// ============================================================================ gather the selected anchors' rows + their logits
template <int FMT>
__global__ __launch_bounds__(256) void rt_gather_kernel(RtLevels enc, int C, int nq, const int* __restrict__ idx, float* __restrict__ embed, float* __restrict__ anchors) {
  const int m = blockIdx.x;                        // image * nq + query
  const int n = m / nq;
  const int a = idx[m];
  int lv, y, x;
  level_of(enc, a, lv, y, x);
  const size_t e = (((size_t)n * enc.h[lv] + y) * enc.w[lv] + x) * enc.cstride[lv] + enc.coff[lv];
  for (int c = threadIdx.x; c < C; c += 256) embed[(size_t)m * C + c] = load1<FMT>(enc.ptr[lv], e + c);
  if (threadIdx.x < 4) {
    const int h = enc.h[lv], w = enc.w[lv];
    const float v = threadIdx.x == 0 ? ((float)x + 0.5f) / (float)w : threadIdx.x == 1 ? ((float)y + 0.5f) / (float)h : 0.05f * (float)(1 << lv);
    anchors[(size_t)m * 4 + threadIdx.x] = anchor_valid(lv, y, x, h, w) ? logf(v / (1.f - v)) : INFINITY;
  }
}

// ============================================================================ reference boxes
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__global__ __launch_bounds__(256) void rt_refer_kernel(const float* __restrict__ delta, int ldd, const float* __restrict__ anchors, float* __restrict__ refer, int M, int mode) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= M * 4) return;
  const int m = t >> 2, j = t & 3;
  const float d = delta[(size_t)m * ldd + j];
  float base;
  if (mode == 0) {
    base = anchors[(size_t)m * 4 + j];
  } else {
    float x = refer[(size_t)m * 16 + j];
    x = fminf(fmaxf(x, 0.f), 1.f);
    base = logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));        // inverse_sigmoid(eps = 1e-5)
  }
  refer[(size_t)m * 16 + j] = sigmoid_f(d + base);
}

// ============================================================================ multi-scale deformable attention sampling
// One workgroup per query, one thread per channel (head = channel / d). Every thread of a head computes the head's 12-way
// softmax and its sampling positions (cheap next to the 48 gathers).
template <int FMT>
__global__ void rt_deform_kernel(RtLevels val, int hd, int nh, int npts, const float* __restrict__ offaw, const float* __restrict__ refer, int nq, float* __restrict__ out) {
  const int m = blockIdx.x, n = m / nq, ch = threadIdx.x;
  if (ch >= hd) return;
  const int d = hd / nh, head = ch / d, L = val.n_levels, LP = L * npts;
  const float* off = offaw + (size_t)m * (nh * LP * 3) + (size_t)head * LP * 2;
  const float* awl = offaw + (size_t)m * (nh * LP * 3) + (size_t)nh * LP * 2 + (size_t)head * LP;
  float mx = awl[0];
  for (int i = 1; i < LP; ++i) mx = fmaxf(mx, awl[i]);
  float den = 0.f;
  for (int i = 0; i < LP; ++i) den += expf(awl[i] - mx);
  const float rx = refer[(size_t)m * 16 + 0], ry = refer[(size_t)m * 16 + 1], rw = refer[(size_t)m * 16 + 2], rh = refer[(size_t)m * 16 + 3];
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = val.h[l], W = val.w[l];
    const size_t img = (size_t)n * H * W;
    for (int p = 0; p < npts; ++p) {
      const int i = l * npts + p;
      const float aw = expf(awl[i] - mx) / den;
      // loc = refer_xy + off / n_points * refer_wh * 0.5; grid = 2 loc - 1; pixel = ((grid + 1) * size - 1) / 2 (align_corners = False)
      const float lx = rx + off[i * 2 + 0] / (float)npts * rw * 0.5f, ly = ry + off[i * 2 + 1] / (float)npts * rh * 0.5f;
      const float gx = 2.f * lx - 1.f, gy = 2.f * ly - 1.f;
      const float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float tx = ix - fx, ty = iy - fy;
      float sv = 0.f;
      auto tap = [&](int yy, int xx, float wgt) {
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          sv += load1<FMT>(val.ptr[l], (img + (size_t)yy * W + xx) * val.cstride[l] + val.coff[l] + ch) * wgt;
      };
      if (ix > -1.f && iy > -1.f && ix < (float)W && iy < (float)H) {   // otherwise all four taps fall outside (also guards the int casts)
        tap(y0, x0, (1.f - tx) * (1.f - ty));
        tap(y0, x0 + 1, tx * (1.f - ty));
        tap(y0 + 1, x0, (1.f - tx) * ty);
        tap(y0 + 1, x0 + 1, tx * ty);
      }
      acc += sv * aw;
    }
  }
  out[(size_t)m * hd + ch] = acc;
}

// ============================================================================ scores, filter, order, scale
__global__ __launch_bounds__(512) void rt_post_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ refer, int nq, int nc, float conf,
                                                      unsigned long long mask0, unsigned long long mask1, float fw, float fh, int max_det,
                                                      float* __restrict__ out_rows, int* __restrict__ out_n, float* __restrict__ raw) {
  __shared__ unsigned long long items[512];
  __shared__ int s_kept;
  const int n = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_kept = 0;
  __syncthreads();
  unsigned long long it = 0ull;
  float best = 0.f;
  int cls = 0;
  if (tid < nq) {
    const size_t m = (size_t)n * nq + tid;
    best = -1.f;
    for (int c = 0; c < nc; ++c) {
      const float s = sigmoid_f(logits[m * ldl + c]);
      if (raw) raw[m * (4 + nc) + 4 + c] = s;
      if (s > best) { best = s; cls = c; }          // first maximum
    }
    if (raw)
      for (int j = 0; j < 4; ++j) raw[m * (4 + nc) + j] = refer[m * 16 + j];
    const bool cls_ok = cls < 64 ? (mask0 >> cls) & 1ull : (mask1 >> (cls - 64)) & 1ull;
    if (best > conf && cls_ok) {
      it = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)tid);   // scores are positive: their bits order them
      atomicAdd(&s_kept, 1);
    }
  }
  items[tid] = it;
  __syncthreads();
  for (int k = 2; k <= 512; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = tid ^ j;
      if (ixj > tid) {
        const unsigned long long a = items[tid], b = items[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? a < b : a > b) { items[tid] = b; items[ixj] = a; }
      }
      __syncthreads();
    }
  const int kept = min(s_kept, max_det);
  if (tid == 0) out_n[n] = kept;
  if (tid < kept) {
    const unsigned long long v = items[tid];
    const int q = (int)(0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull));
    const size_t m = (size_t)n * nq + q;
    const float cx = refer[m * 16 + 0], cy = refer[m * 16 + 1], hw = refer[m * 16 + 2] / 2.f, hh = refer[m * 16 + 3] / 2.f;
    // the class again (cheap) -- the sort carried the score and the query only
    float b2 = -1.f;
    int c2 = 0;
    for (int c = 0; c < nc; ++c) {
      const float s = sigmoid_f(logits[m * ldl + c]);
      if (s > b2) { b2 = s; c2 = c; }
    }
    float* o = out_rows + ((size_t)n * max_det + tid) * 6;
    o[0] = (cx - hw) * fw; o[1] = (cy - hh) * fh; o[2] = (cx + hw) * fw; o[3] = (cy + hh) * fh;   // xywh2xyxy, then x frame width / height
    o[4] = b2;
    o[5] = (float)c2;
  }
}

inline unsigned blocks_for(size_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

// ---------------------------------------------------------------------------- launchers
void launch_rt_stem1(int fmt, const void* img, int n, int H, int W, const float* w27, const float* bias, const RtMap& out, int* sat, hipStream_t s) {
  GTX_CHECK(out.c % 16 == 0 && out.h * 2 == H && out.w * 2 == W, "rt_stem1: bad shapes");
  const size_t total = (size_t)n * out.h * out.w;
  RT_FMT(fmt, hipLaunchKernelGGL(rt_stem1_kernel<F>, dim3(blocks_for(total)), dim3(256), 0, s, (const uchar4*)img, n, H, W, w27, bias, out, sat));
}

void launch_rt_pool2(int fmt, const RtMap& in, const RtMap& out, int n, int* sat, hipStream_t s) {
  GTX_CHECK(in.c % 8 == 0 && in.c == out.c && in.h == out.h && in.w == out.w, "rt_pool2: bad shapes");
  const size_t total = (size_t)n * in.h * in.w * (in.c / 8);
  RT_FMT(fmt, hipLaunchKernelGGL(rt_pool2_kernel<F>, dim3(blocks_for(total)), dim3(256), 0, s, in, out, n, sat));
}

void launch_rt_dwconv(int fmt, const RtMap& in, const RtMap& out, int n, int k, int stride, const float* w, const float* bias, int act, int* sat, hipStream_t s) {
  GTX_CHECK(in.c % 8 == 0 && in.c == out.c && (k == 3 || k == 5) && (stride == 1 || stride == 2), "rt_dwconv: unsupported %dx%d stride %d on %d channels", k, k, stride, in.c);
  GTX_CHECK(out.h == (in.h + 2 * (k / 2) - k) / stride + 1 && out.w == (in.w + 2 * (k / 2) - k) / stride + 1, "rt_dwconv: output size");
  const size_t total = (size_t)n * out.h * out.w * (in.c / 8);
  static const bool tiled = [] { const char* e = getenv("GTX_RT_DW_TILE"); return !(e && e[0] == '0'); }();
  if (stride == 1 && in.c % 32 == 0 && tiled) {
    const dim3 grid(cdiv(out.w, 8) * cdiv(out.h, 8), in.c / 32, n);
    if (k == 3) RT_FMT(fmt, hipLaunchKernelGGL((rt_dwconv_tile_kernel<F, 3>), grid, dim3(256), 0, s, in, out, w, bias, act, sat));
    else RT_FMT(fmt, hipLaunchKernelGGL((rt_dwconv_tile_kernel<F, 5>), grid, dim3(256), 0, s, in, out, w, bias, act, sat));
    return;
  }
  if (k == 3) RT_FMT(fmt, hipLaunchKernelGGL((rt_dwconv_kernel<F, 3>), dim3(blocks_for(total)), dim3(256), 0, s, in, out, n, stride, w, bias, act, sat));
  else RT_FMT(fmt, hipLaunchKernelGGL((rt_dwconv_kernel<F, 5>), dim3(blocks_for(total)), dim3(256), 0, s, in, out, n, stride, w, bias, act, sat));
}

void launch_rt_upsample2x(int fmt, const RtMap& in, const RtMap& out, int n, hipStream_t s) {
  GTX_CHECK(in.c % 8 == 0 && in.c == out.c && out.h == 2 * in.h && out.w == 2 * in.w, "rt_upsample2x: bad shapes");
  const size_t total = (size_t)n * out.h * out.w * (in.c / 8);
  hipLaunchKernelGGL(rt_upsample2x_kernel, dim3(blocks_for(total)), dim3(256), 0, s, in, out, n, 8 * fmt_size(fmt));
  GTX_HIP(hipGetLastError());
}

void launch_rt_tokens_in(int fmt, const RtMap& in, int n, const float* pos, float* src, float* q, hipStream_t s) {
  GTX_CHECK(in.c % 8 == 0, "rt_tokens_in: channels");
  const size_t total = (size_t)n * in.h * in.w * (in.c / 8);
  RT_FMT(fmt, hipLaunchKernelGGL(rt_tokens_in_kernel<F>, dim3(blocks_for(total)), dim3(256), 0, s, in, n, pos, src, q));
}

void launch_rt_mask_invalid(int fmt, const RtMap& m, int n, int level, hipStream_t s) {
  const size_t total = (size_t)n * m.h * m.w * (m.c / 8);
  hipLaunchKernelGGL(rt_mask_invalid_kernel, dim3(blocks_for(total)), dim3(256), 0, s, m, n, level, 8 * fmt_size(fmt));
  GTX_HIP(hipGetLastError());
}

void launch_rt_linear(const RtLinear& p, hipStream_t s) {
  GTX_CHECK(p.K % 16 == 0 && p.Nout % 16 == 0 && p.M > 0 && p.ldx % 4 == 0 && (!p.x2 || p.ldx2 % 4 == 0), "rt_linear: K=%d Nout=%d M=%d", p.K, p.Nout, p.M);
  GTX_CHECK(!p.x2 || p.x2_cols % 64 == 0 || p.x2_cols >= p.Nout, "rt_linear: the second addend feeds %d columns (a multiple of 64, or all %d)", p.x2_cols, p.Nout);
  hipLaunchKernelGGL(rt_linear_kernel, dim3(cdiv(p.M, 16), cdiv(p.Nout, 64)), dim3(256), 0, s, p);
  GTX_HIP(hipGetLastError());
}

void launch_rt_layernorm(const RtRows& in, const RtRows& out, long rows, int C, const float* gamma, const float* beta, int* sat, hipStream_t s) {
  GTX_CHECK(C % 8 == 0 && C <= 1024, "rt_layernorm: C=%d", C);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define RT_LN(FI, FO) hipLaunchKernelGGL((rt_layernorm_kernel<FI, FO>), grid, block, 0, s, in, out, rows, C, gamma, beta, sat)
  if (in.fmt == DT_F32 && out.fmt == DT_F32) RT_LN(DT_F32, DT_F32);
  else if (in.fmt == DT_F32 && out.fmt == DT_F32S) RT_LN(DT_F32, DT_F32S);
  else if (in.fmt == DT_F32 && out.fmt == DT_F16) RT_LN(DT_F32, DT_F16);
  else if (in.fmt == DT_F32S && out.fmt == DT_F32S) RT_LN(DT_F32S, DT_F32S);
  else if (in.fmt == DT_F16 && out.fmt == DT_F16) RT_LN(DT_F16, DT_F16);
  else fail(-3, "rt_layernorm: formats %d -> %d", in.fmt, out.fmt);
#undef RT_LN
  GTX_HIP(hipGetLastError());
}

void launch_rt_mha(const float* qkv, int ld, int n, int T, int C, int heads, float* out, int ldo, hipStream_t s) {
  const int d = C / heads;
  GTX_CHECK(C % heads == 0 && ld % 4 == 0 && ldo % 4 == 0, "rt_mha: C=%d heads=%d", C, heads);
  const dim3 grid(cdiv(T, 64), heads, n), block(256);
  static const bool mfma = [] { const char* e = getenv("GTX_RT_MHA_MFMA"); return !(e && e[0] == '0'); }();
  if (d == 32 && mfma) hipLaunchKernelGGL(rt_mha32_kernel, grid, block, 0, s, qkv, ld, T, C, out, ldo);
  else if (d == 32) hipLaunchKernelGGL(rt_mha_kernel<32>, grid, block, 0, s, qkv, ld, T, C, out, ldo);
  else if (d == 16) hipLaunchKernelGGL(rt_mha_kernel<16>, grid, block, 0, s, qkv, ld, T, C, out, ldo);
  else if (d == 8) hipLaunchKernelGGL(rt_mha_kernel<8>, grid, block, 0, s, qkv, ld, T, C, out, ldo);
  else fail(-3, "rt_mha: head dimension %d is not built (8, 16, 32)", d);
  GTX_HIP(hipGetLastError());
}

void launch_rt_topk(int fmt, const RtLevels& scores, int nc, int n, int nq, unsigned* keys_scratch, int* out_idx, hipStream_t s) {
  int S = 0;
  for (int l = 0; l < scores.n_levels; ++l) S += scores.h[l] * scores.w[l];
  GTX_CHECK(nq >= 1 && nq <= 1024 && S >= nq, "rt_topk: %d queries of %d anchors", nq, S);
  if (fmt == DT_F16) hipLaunchKernelGGL(rt_topk_keys_kernel<DT_F16>, dim3(cdiv(S, 256), n), dim3(256), 0, s, scores, nc, S, keys_scratch);
  else hipLaunchKernelGGL(rt_topk_keys_kernel<DT_F32>, dim3(cdiv(S, 256), n), dim3(256), 0, s, scores, nc, S, keys_scratch);   // plain fp32 maps (both fp32-grade paths)
  hipLaunchKernelGGL(rt_topk_kernel, dim3(n), dim3(1024), 0, s, S, nq, (const unsigned*)keys_scratch, out_idx);
  GTX_HIP(hipGetLastError());
}

void launch_rt_gather(int fmt, const RtLevels& enc, int C, int n, int nq, const int* idx, float* embed, float* anchors, hipStream_t s) {
  RT_FMT(fmt, hipLaunchKernelGGL(rt_gather_kernel<F>, dim3(n * nq), dim3(256), 0, s, enc, C, nq, idx, embed, anchors));
}

void launch_rt_refer(const float* delta, int ldd, const float* anchors, float* refer, int M, int mode, hipStream_t s) {
  hipLaunchKernelGGL(rt_refer_kernel, dim3(cdiv(M * 4, 256)), dim3(256), 0, s, delta, ldd, anchors, refer, M, mode);
  GTX_HIP(hipGetLastError());
}

void launch_rt_deform(int fmt, const RtLevels& value, int hd, int nh, int npts, const float* offaw, const float* refer, int n, int nq, float* out, hipStream_t s) {
  GTX_CHECK(hd % nh == 0 && hd <= 1024, "rt_deform: hd=%d heads=%d", hd, nh);
  RT_FMT(fmt, hipLaunchKernelGGL(rt_deform_kernel<F>, dim3(n * nq), dim3((hd + 63) / 64 * 64), 0, s, value, hd, nh, npts, offaw, refer, nq, out));
}

void launch_rt_post(const float* logits, int ldl, const float* refer, int n, int nq, int nc, float conf, const unsigned long long class_mask[2],
                    int frame_w, int frame_h, int max_det, float* out_rows, int* out_n, float* raw, hipStream_t s) {
  GTX_CHECK(nq <= 512, "rt_post: %d queries (at most 512)", nq);
  hipLaunchKernelGGL(rt_post_kernel, dim3(n), dim3(512), 0, s, logits, ldl, refer, nq, nc, conf, class_mask[0], class_mask[1], (float)frame_w, (float)frame_h,
                     max_det, out_rows, out_n, raw);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
