// The Detect box branch at candidate anchors only (default fp32 path, split-f16x3). gfx950.
//
// Detect's box branch is two 3x3 convolutions (cv2[l][0]: Cin -> 64, cv2[l][1]: 64 -> 64) whose output is read by the decode at
// the anchors that pass the score gate and nowhere else: a few hundred to a few thousand of the 75 600 anchors of a 1920 x 1920
// input. This kernel evaluates the two layers for those anchors alone -- a wave takes four candidates of one level: the 3 x 3
// neighbourhood of cv2[l][0] outputs the second layer needs (nine MFMA columns per candidate), then cv2[l][1] at the anchors --
// with the arithmetic of the dense
// kernels it replaces (conv_k32_split.hip: same packed weight images and power-of-two scales, same v_mfma_f32_16x16x32_f16
// sequence per output pixel -- chunk, kernel row, kernel column, small terms first -- same bias start, SiLU and hi / lo split,
// zero padding at the image border for both layers), so a candidate's 64 box features are the dense path's bit for bit.
// The wave then decodes its candidates' boxes (the final 1x1 convolution + DFL of head_boxes_kernel, same operations in the same
// order), so the sparse path is one launch between the score gate and the NMS.
// The dense convolutions stay available (Detector: GTX_SPARSE_BOX=0, the debug read-backs, more candidates than the buffer holds).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "det_kernels.hpp"

namespace gtx {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#define GTXH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// conv_k32_split.hip's epilogue arithmetic (same operations in the same order)
__device__ __forceinline__ float2v silu2(const float2v v) {
  const float2v t = v * -1.44269504088896341f;
  const float2v d = float2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.f;
  return v * float2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ void split2(const float2v v, unsigned& hi, unsigned& lo, bool& sat) {
  const float2v x = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
  sat |= x.x != v.x || x.y != v.y;
  const half2v h = __builtin_convertvector(x, half2v);
  const half2v l = __builtin_convertvector(x - __builtin_convertvector(h, float2v), half2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

constexpr int kRowBytes = 256;                    // 64 channels in pair format
constexpr int kCandPerWave = 4;                   // candidates of one level a wave works on: they share every weight fragment
constexpr int kWaveLds = kCandPerWave * 9 * kRowBytes;   // their first-layer pixels (3 x 3 each)

// packed weight image of a 64-cout tile with 32-channel chunks (pack_conv_weights_split): [chunk][tap][n 64][8 swizzled 16-B chunks]
__device__ __forceinline__ const char* wrow(const void* w, int step /* chunk * 9 + tap */, int row) {
  return static_cast<const char*>(w) + ((size_t)step * 64 + row) * 128;
}

// grid.x = levels x groups: workgroup g of level l takes candidates [16 g, 16 g + 16) of the level's list (head_candidates_kernel
// files every candidate under its level), four per wave. A wave's candidates share the weights: 8 fragment loads per K step feed
// 48 MFMAs, and the four independent accumulator sets hide the loads' latency.
__global__ __launch_bounds__(256) void head_sparse_box_kernel(const SparseBox sb, const NmsBuffers nb, int groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.y;
  const int l = blockIdx.x / groups, g = blockIdx.x - l * groups;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cnt = min(nb.lvl_count[n * kMaxLevels + l], nb.lvl_cap);
  const int first = (g * 4 + wave) * kCandPerWave;
  if (first >= cnt) return;                       // wave-uniform; no workgroup barrier below
  const SparseBoxLevel& L = sb.lv[l];
  const int col = lane & 15, kg = lane >> 4;
  const int pos = min(col, 8);                    // first-layer pixel of this MFMA column (columns 9..15 repeat the last one)
  int ci[kCandPerWave], y1[kCandPerWave], x1[kCandPerWave];
  bool valid1[kCandPerWave];
#pragma unroll
  for (int c = 0; c < kCandPerWave; ++c) {
    const int k = min(first + c, cnt - 1);        // a short last group repeats its last candidate (not stored twice)
    ci[c] = __builtin_amdgcn_readfirstlane(nb.lvl_list[((size_t)n * kMaxLevels + l) * nb.lvl_cap + k]);
    const int a = __builtin_amdgcn_readfirstlane(nb.cand_anchor[(size_t)n * nb.cap + ci[c]]);
    const int la = a - L.anchor_begin;
    const int ay = la / L.W, ax = la - ay * L.W;
    y1[c] = ay + pos / 3 - 1;
    x1[c] = ax + pos % 3 - 1;
    valid1[c] = y1[c] >= 0 && y1[c] < L.H && x1[c] >= 0 && x1[c] < L.W;   // outside the map: the second layer's zero padding
  }
  const float* __restrict__ in = static_cast<const float*>(L.in) + (size_t)n * L.H * L.W * L.cstride + L.coff + kg * 8;
  char* lds = smem + wave * kWaveLds;

  // ---- first layer: nine pixels x 64 channels per candidate, K = Cin x 9 ----
  floatx4 acc[kCandPerWave][4];
  {
    const float inv_sc = __builtin_amdgcn_rcpf(L.sc1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(L.b1 + 16 * q + 4 * kg);
#pragma unroll
      for (int c = 0; c < kCandPerWave; ++c) acc[c][q] = floatx4{b.x * inv_sc, b.y * inv_sc, b.z * inv_sc, b.w * inv_sc};
    }
  }
  const int nsteps = (L.cin / 32) * 9;
  // three register sets in rotation: the loads of steps s + 1 and s + 2 are in flight while step s multiplies (a step's loads are
  // L2 round trips; nine taps per chunk: the step count is a multiple of three)
  half8 bh[3][kCandPerWave], bl[3][kCandPerWave], ah[3][4], al[3][4];
#define GTXH_LOAD1(STEP, R)                                                                     \
  {                                                                                             \
    const int ch__ = (STEP) / 9, tap__ = (STEP) - ch__ * 9;                                     \
    _Pragma("unroll") for (int c = 0; c < kCandPerWave; ++c) {                                  \
      const int y__ = y1[c] + tap__ / 3 - 1, x__ = x1[c] + tap__ % 3 - 1;                       \
      uint4 h__ = make_uint4(0, 0, 0, 0), l__ = make_uint4(0, 0, 0, 0);                         \
      if (y__ >= 0 && y__ < L.H && x__ >= 0 && x__ < L.W) {                                     \
        const uint4* s__ = reinterpret_cast<const uint4*>(in + ((size_t)y__ * L.W + x__) * L.cstride + ch__ * 32); \
        h__ = s__[0];                                                                           \
        l__ = s__[1];                                                                           \
      }                                                                                         \
      bh[R][c] = __builtin_bit_cast(half8, h__);                                                \
      bl[R][c] = __builtin_bit_cast(half8, l__);                                                \
    }                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
      const int row__ = 16 * q + col;                                                           \
      const char* w__ = wrow(L.w1, STEP, row__);                                                \
      ah[R][q] = *reinterpret_cast<const half8*>(w__ + ((kg ^ ((row__ >> 1) & 7)) << 4));       \
      al[R][q] = *reinterpret_cast<const half8*>(w__ + (((4 + kg) ^ ((row__ >> 1) & 7)) << 4)); \
    }                                                                                           \
  }
#define GTXH_STEP1(S, R)                                                                        \
  {                                                                                             \
    if ((S) + 2 < nsteps) GTXH_LOAD1((S) + 2, ((R) + 2) % 3)                                    \
    _Pragma("unroll") for (int c = 0; c < kCandPerWave; ++c)                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                           \
        acc[c][q] = GTXH_MFMA(al[R][q], bh[R][c], acc[c][q]);                                   \
        acc[c][q] = GTXH_MFMA(ah[R][q], bl[R][c], acc[c][q]);                                   \
        acc[c][q] = GTXH_MFMA(ah[R][q], bh[R][c], acc[c][q]);                                   \
      }                                                                                         \
  }
  GTXH_LOAD1(0, 0)
  GTXH_LOAD1(1, 1)
  for (int s = 0; s < nsteps; s += 3) {
    GTXH_STEP1(s, 0)
    GTXH_STEP1(s + 1, 1)
    GTXH_STEP1(s + 2, 2)
  }
#undef GTXH_STEP1
#undef GTXH_LOAD1
  // SiLU, hi / lo split, pair rows in LDS: lane (col, kg) of block q holds channels 16 q + 4 kg + 0..3 of pixel col; two
  // v_permlane16_swap make the 8-channel group's hi chunk (even kg) and lo chunk (odd kg): byte 64 q + 16 kg of the row
  bool sat = false;
#pragma unroll
  for (int c = 0; c < kCandPerWave; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float2v v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        v[e] = silu2(float2v{acc[c][q][2 * e], acc[c][q][2 * e + 1]} * L.sc1);
        if (!valid1[c]) v[e] = float2v{0.f, 0.f};
      }
      uint2 hi, lo;
      bool s1 = false;
      split2(v[0], hi.x, lo.x, s1);
      split2(v[1], hi.y, lo.y, s1);
      sat |= s1 && col < 9;
      const auto sx = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
      if (col < 9) *reinterpret_cast<uint4*>(lds + (c * 9 + col) * kRowBytes + 64 * q + 16 * kg) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    }

  // ---- second layer at the anchors: K = 64 x 9 over each candidate's nine staged pixels; MFMA column j works for candidate j & 3 ----
  floatx4 acc2[4];
  {
    const float inv_sc = __builtin_amdgcn_rcpf(L.sc2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b = *reinterpret_cast<const float4*>(L.b2 + 16 * q + 4 * kg);
      acc2[q] = floatx4{b.x * inv_sc, b.y * inv_sc, b.z * inv_sc, b.w * inv_sc};
    }
  }
  const char* mine = lds + (col & 3) * 9 * kRowBytes;
  half8 wh[3][4], wl[3][4];
#define GTXH_LOAD2(STEP, R)                                                                     \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
    const int row__ = 16 * q + col;                                                             \
    const char* w__ = wrow(L.w2, STEP, row__);                                                  \
    wh[R][q] = *reinterpret_cast<const half8*>(w__ + ((kg ^ ((row__ >> 1) & 7)) << 4));         \
    wl[R][q] = *reinterpret_cast<const half8*>(w__ + (((4 + kg) ^ ((row__ >> 1) & 7)) << 4));   \
  }
#define GTXH_STEP2(S, R)                                                                        \
  {                                                                                             \
    if ((S) + 2 < 18) GTXH_LOAD2((S) + 2, ((R) + 2) % 3)                                        \
    const int ch__ = (S) / 9, tap__ = (S) - ch__ * 9;                                           \
    const char* r__ = mine + tap__ * kRowBytes + (4 * ch__ + kg) * 32;   /* the tap's pixel, 8-channel group 4 ch + kg: hi chunk, lo chunk */ \
    const half8 xh__ = *reinterpret_cast<const half8*>(r__), xl__ = *reinterpret_cast<const half8*>(r__ + 16); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
      acc2[q] = GTXH_MFMA(wl[R][q], xh__, acc2[q]);                                             \
      acc2[q] = GTXH_MFMA(wh[R][q], xl__, acc2[q]);                                             \
      acc2[q] = GTXH_MFMA(wh[R][q], xh__, acc2[q]);                                             \
    }                                                                                           \
  }
  GTXH_LOAD2(0, 0)
  GTXH_LOAD2(1, 1)
#pragma unroll 1
  for (int s = 0; s < 18; s += 3) {
    GTXH_STEP2(s, 0)
    GTXH_STEP2(s + 1, 1)
    GTXH_STEP2(s + 2, 2)
  }
#undef GTXH_STEP2
#undef GTXH_LOAD2
  // the candidates' 64 box features (plain fp32: ConvProblem::out_plain of the dense layer) -> LDS, where the decode below reads
  // them; the first-layer rows are no longer needed (a wave's LDS operations execute in order)
  float* fl = reinterpret_cast<float*>(lds);       // [candidate][64]
  if (col < kCandPerWave) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2v v0 = silu2(float2v{acc2[q][0], acc2[q][1]} * L.sc2), v1 = silu2(float2v{acc2[q][2], acc2[q][3]} * L.sc2);
      *reinterpret_cast<float4*>(fl + col * 64 + 16 * q + 4 * kg) = make_float4(v0.x, v0.y, v1.x, v1.y);
    }
  }
  // ---- box decode of the wave's candidates (head_boxes_kernel's arithmetic: the 64 x 64 final 1x1 convolution as one fmaf chain
  // per output in feature order, DFL softmax expectation per side, dist2bbox, xywh -> xyxy) ----
#pragma unroll 1
  for (int c = 0; c < kCandPerWave; ++c) {
    if (first + c >= cnt) break;                   // wave-uniform
    float a = L.bb[lane];
    const float* wr = L.wb + lane;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) a = fmaf(fl[c * 64 + k], wr[(size_t)k * 64], a);
    const int anchor = __builtin_amdgcn_readfirstlane(nb.cand_anchor[(size_t)n * nb.cap + ci[c]]);
    const int la = anchor - L.anchor_begin;
    // softmax over the 16 lanes of a side
    float m = a;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const float e = expf(a - m);
    float se = e, sw = e * (float)(lane & 15);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
      se += __shfl_xor(se, o, 64);
      sw += __shfl_xor(sw, o, 64);
    }
    const float d = sw / se;
    const float d0 = __shfl(d, 0, 64), d1 = __shfl(d, 16, 64), d2 = __shfl(d, 32, 64), d3 = __shfl(d, 48, 64);
    const float ax = (float)(la % L.W) + 0.5f, ay = (float)(la / L.W) + 0.5f;
    const float bx1 = ax - d0, by1 = ay - d1, bx2 = ax + d2, by2 = ay + d3;
    const float4 b = make_float4((bx1 + bx2) * 0.5f * L.stride, (by1 + by2) * 0.5f * L.stride, (bx2 - bx1) * L.stride, (by2 - by1) * L.stride);
    if (lane == 0) {
      const float hw = b.z / 2.f, hh = b.w / 2.f;   // xywh2xyxy
      reinterpret_cast<float4*>(nb.cand_box)[(size_t)n * nb.cap + ci[c]] = make_float4(b.x - hw, b.y - hh, b.x + hw, b.y + hh);
    }
  }
  if (sb.sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(sb.sat_flag, 1);
}

}  // namespace

void launch_head_sparse_box(const SparseBox& sb, int n, const NmsBuffers& nb, hipStream_t s) {
  GTX_CHECK(sb.cap > 0 && nb.lvl_count != nullptr && nb.lvl_list != nullptr && nb.lvl_cap == sb.cap, "sparse box branch: no buffers");
  const int groups = (nb.lvl_cap + 4 * kCandPerWave - 1) / (4 * kCandPerWave);
  hipLaunchKernelGGL(head_sparse_box_kernel, dim3(sb.n_levels * groups, n), dim3(256), 4 * kWaveLds, s, sb, nb, groups);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
