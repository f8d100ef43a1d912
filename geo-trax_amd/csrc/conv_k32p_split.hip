// 1x1 (pointwise) split-f16x3 convolution on v_mfma_f32_16x16x32_f16 ("K32 pointwise forms", ConvConfig::variant 6). gfx950 only.
// EXPERIMENT, closed with numbers: built by `make K32P=1` only, selected with GTX_K32P=1 / 2, never by default.
//
// The arithmetic contract, the packed weight image (pack_conv_weights_split, 32-channel chunks, one tap) and the output tile
// (8 x 16 pixels x 64 couts per 4-wave workgroup, wave w owning tile rows 2w and 2w + 1) are conv_k32_split.hip's; what differs
// is the K loop. A 1x1 layer has no taps to reuse a staged patch over, so a stage here is TWO 32-channel chunks: 128 pixels x
// 256 B of activations (32 KB: every pixel's 256-byte run of the pair format is one coalesced read) beside 2 x 64 couts x 128 B
// of weights (16 KB) -- 48 KB per workgroup, three per CU -- and 48 MFMAs per wave between two barrier pairs. Fragments are read
// as in the 3x3 kernel: the weights' (A) one 16-cout block ahead, the second chunk's pixels (B) while the first chunk multiplies.
// The question it answers: RT-DETR's HGNetv2 is mostly such layers (57 launches of a pass, K up to 3328: 35 % of its time on the
// 32x32x16 kernel at 200 TFLOP/s, where the 3x3 layers reach 350 on the 16x16x32 instruction) -- is it the instruction shape?
// It is not (MI355X, 3840x2160 -> 1920^2, batch 2; profiles/r06_k32p_probe.txt): 4.04 ms for the 57 launches with the pixels
// staged (201 TFLOP/s), 4.86 ms with the pixels loaded straight into the operand registers (two half-used 128-byte lines per
// lane and instruction), 3.92 ms on the 32x32x16 kernel. A 1x1 layer stages 9 x the bytes per MFMA of a 3x3 layer (no taps to
// reuse the patch over): per 32-channel chunk a workgroup moves 24 KB through the CU's load path and 72 KB through LDS for 384
// MFMA cycles per SIMD, whichever instruction issues them. Not for launches with a second (upsampled) source or a fused stage:
// those stay on the 32x32x16 kernel (conv_igemm_split.hip), which reads the same weight image.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdlib>
#include <mutex>

#include "conv_igemm.hpp"

namespace gtx {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#define GTXP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

struct K32PTile {
  static constexpr int TH = 8, TW = 16, BN = 64, KC = 32, CPR = 4, RB = 128, CS = 2;   // CS chunks per stage
  static constexpr int NPIX = TH * TW;
  static constexpr int PATCH_CHUNK_BYTES = NPIX * RB;                 // 16 KB
  static constexpr int PATCH_UNITS = CS * NPIX * CPR;                 // one unit = 8 channels of one pixel (hi chunk, lo chunk): 1024
  static constexpr int PATCH_SLOTS = PATCH_UNITS / 256;               // 4
  static constexpr int PATCH_BYTES = CS * PATCH_CHUNK_BYTES;
  static constexpr int W_CHUNK_U4 = BN * 8;                           // 16-byte pieces of one chunk's weights: 512
  static constexpr int W_SLOTS = CS * W_CHUNK_U4 / 256;               // 4
  static constexpr int W_BYTES = CS * W_CHUNK_U4 * 16;
  static constexpr int STAGE_BYTES = PATCH_BYTES + W_BYTES;
  static constexpr int EPI_PITCH = BN * 4 + 16;
  static constexpr int EPI_BYTES = 4 * 32 * EPI_PITCH;
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int LDS_DIRECT_BYTES = W_BYTES > EPI_BYTES ? W_BYTES : EPI_BYTES;     // direct form: only the weights are staged
  static __host__ __device__ constexpr int swz(int row) { return (row >> 1) & 7; }
};
static_assert(3 * K32PTile::LDS_BYTES <= 160 * 1024, "three workgroups per CU");
static_assert(K32PTile::PATCH_SLOTS == 4 && K32PTile::W_SLOTS == 4, "four slots each");

// conv_igemm_split.hip's epilogue arithmetic (same operations in the same order)
__device__ __forceinline__ float2v relu2(const float2v v) { return float2v{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)}; }
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

// DIRECT: the pixels' fragments never pass through LDS. A wave's pixels are its own (rows 2w, 2w + 1 of the tile), and lane (col, kg)'s
// operand of chunk ch is exactly the 32-byte unit 4 ch + kg of pixel (row, col) in the pair format: the four (row, chunk) units a
// lane needs per stage are prefetched straight into the registers the MFMAs read; only the weights (shared by the four waves) are staged.
template <bool DIRECT>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 256), amdgpu_waves_per_eu(3)))
void conv_k32p_split_kernel(const ConvGroup g) {
  using Tile = K32PTile;
  constexpr int RB = Tile::RB, BN = Tile::BN, CPR = Tile::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = DIRECT ? smem : smem + Tile::PATCH_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // launch header and XCD-aware logical block id: conv_igemm_split.hip
  const int cnt = g.count;
  int bb[kMaxGroup];
#pragma unroll
  for (int i = 0; i < kMaxGroup; ++i) bb[i] = g.p[i].block_begin;
  const int xcd = blockIdx.x & 7;
  const int L = g.xcd_begin[xcd] + (int)(blockIdx.x >> 3);
  if (L >= g.xcd_begin[xcd + 1]) return;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < cnt && L >= bb[i]) pi = i;
  const ConvProblem P = g.p[pi];

  const int lb = L - P.block_begin;
  const int ct = lb % P.n_ct;
  const int pt = lb / P.n_ct;
  const int tx = pt % P.tiles_x;
  const int t2 = pt / P.tiles_x;
  const int ty = t2 % P.tiles_y + P.ty_first;
  const int n = t2 / P.tiles_y;
  const int oy0 = ty * Tile::TH, ox0 = tx * Tile::TW;

  const float* __restrict__ in = static_cast<const float*>(P.in);
  const int nchunks = P.Cin / Tile::KC;
  const int nstages = (nchunks + 1) >> 1;

  const int col = lane & 15, kg = lane >> 4;
  const int p0 = (2 * wave) * 16 + col;           // tile pixel of (row 2 wave, column col)

  // staged form: unit qid = tid + 256 s is pixel qid / 8 of the tile, unit qid % 8 of the stage's 64 channels (chunk (qid % 8) / 4).
  // direct form: slot s = 2 chunk + m is unit 4 chunk + kg of this lane's pixel (row 2 wave + m, column col)
  int goff[Tile::PATCH_SLOTS];                    // element offset of the unit, -1 = zero fill (pixel outside the map)
  int loff[Tile::PATCH_SLOTS];                    // LDS byte offset of the unit's hi chunk; its lo chunk: ^ 64
#pragma unroll
  for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {
    if (DIRECT) {
      const int oy = oy0 + 2 * wave + (s & 1), ox = ox0 + col;
      goff[s] = (oy < P.H && ox < P.W) ? ((n * P.H + oy) * P.W + ox) * P.in_cstride + P.in_coff + (4 * (s >> 1) + kg) * 8 : -1;
      loff[s] = 0;
    } else {
      const int qid = tid + 256 * s;
      const int p = qid >> 3, c = qid & 7;
      const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
      const bool inb = oy < P.H && ox < P.W;
      goff[s] = inb ? ((n * P.H + oy) * P.W + ox) * P.in_cstride + P.in_coff + c * 8 : -1;
      loff[s] = (c >> 2) * Tile::PATCH_CHUNK_BYTES + p * RB + (((c & 3) ^ Tile::swz(p)) << 4);
    }
  }
  // packed image: [cout tile][chunk][n][8 swizzled 16-byte pieces]: a stage's two chunks are 1024 contiguous uint4
  const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>(P.wpack) + (size_t)ct * nchunks * Tile::W_CHUNK_U4 + tid;

  uint4 pre_a[Tile::PATCH_SLOTS], pre_b[Tile::PATCH_SLOTS];
  uint4 pw0, pw1, pw2, pw3;
  // a stage's second chunk does not exist when Cin / 32 is odd and the stage is the last: its units (slots whose unit index has
  // bit 2 set = odd qid >> 2 ... c >= 4) and its weights (pieces 512..1023) are zero-filled, its MFMAs skipped
#define GTXP_PREFETCH(STAGE)                                                                 \
  {                                                                                          \
    const int c0__ = (STAGE) * (Tile::CS * Tile::KC);                                        \
    const bool two__ = 2 * (STAGE) + 1 < nchunks;                                            \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      uint4 va__ = make_uint4(0, 0, 0, 0), vb__ = make_uint4(0, 0, 0, 0);                    \
      if (goff[s] >= 0 && (two__ || (DIRECT ? s < 2 : ((tid + 256 * s) & 4) == 0))) {        \
        const uint4* src__ = reinterpret_cast<const uint4*>(in + goff[s] + c0__);            \
        va__ = src__[0];                                                                     \
        vb__ = src__[1];                                                                     \
      }                                                                                      \
      pre_a[s] = va__;                                                                       \
      pre_b[s] = vb__;                                                                       \
    }                                                                                        \
    const uint4* w__ = wsrc + (size_t)(STAGE) * (Tile::CS * Tile::W_CHUNK_U4);               \
    pw0 = w__[0]; pw1 = w__[256];                                                            \
    if (two__) { pw2 = w__[512]; pw3 = w__[768]; }                                           \
  }
#define GTXP_COMMIT()                                                                        \
  {                                                                                          \
    if (!DIRECT) {                                                                           \
      _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                        \
        *reinterpret_cast<uint4*>(lds_patch + loff[s]) = pre_a[s];                           \
        *reinterpret_cast<uint4*>(lds_patch + (loff[s] ^ (CPR << 4))) = pre_b[s];            \
      }                                                                                      \
    }                                                                                        \
    uint4* d__ = reinterpret_cast<uint4*>(lds_w) + tid;                                      \
    d__[0] = pw0; d__[256] = pw1; d__[512] = pw2; d__[768] = pw3;                            \
  }

  pw2 = pw3 = make_uint4(0, 0, 0, 0);
  GTXP_PREFETCH(0)

  // accumulators start at bias / acc_scale (conv_igemm_split.hip): lane (col, kg) of block a holds couts 16 a + 4 kg + 0..3
  floatx4 acc[4][2];
  {
    const float inv_sc = __builtin_amdgcn_rcpf(P.acc_scale);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (P.bias) b = *reinterpret_cast<const float4*>(P.bias + ct * BN + 16 * a + 4 * kg);
#pragma unroll
      for (int m = 0; m < 2; ++m) acc[a][m] = floatx4{b.x * inv_sc, b.y * inv_sc, b.z * inv_sc, b.w * inv_sc};
    }
  }

  half8 bh[2][2], bl[2][2];                      // [slot = chunk of the stage][pixel block m]
  half8 ah[2], al[2];                            // [slot]
  // pixels' fragments of chunk CH of the stage -> slot CH; piece Q = 0..3 is one 16-byte read (m = Q >> 1, hi / lo = Q & 1)
#define GTXP_LOAD_B(CH, Q)                                                                     \
    {                                                                                          \
      const int p__ = p0 + ((Q) >> 1) * 16;                                                    \
      const char* pr__ = lds_patch + (CH) * Tile::PATCH_CHUNK_BYTES + p__ * RB;                \
      if (((Q) & 1) == 0) bh[CH][(Q) >> 1] = *reinterpret_cast<const half8*>(pr__ + ((kg ^ Tile::swz(p__)) << 4)); \
      else bl[CH][(Q) >> 1] = *reinterpret_cast<const half8*>(pr__ + (((CPR + kg) ^ Tile::swz(p__)) << 4)); \
    }
  // weights' fragments of (chunk CH, cout block A) -> slot
#define GTXP_LOAD_A(CH, A, SLOT)                                                               \
    {                                                                                          \
      const int nrow__ = 16 * (A) + col;                                                       \
      const char* wr__ = lds_w + ((CH) * BN + nrow__) * RB;                                    \
      ah[SLOT] = *reinterpret_cast<const half8*>(wr__ + ((kg ^ Tile::swz(nrow__)) << 4));     \
      al[SLOT] = *reinterpret_cast<const half8*>(wr__ + (((CPR + kg) ^ Tile::swz(nrow__)) << 4)); \
    }
  // unit U = 4 CH + a: 6 MFMAs; reads the weights of unit U + 1 and, during the first chunk, one piece of the second chunk's pixels
#define GTXP_UNIT(U)                                                                           \
    {                                                                                          \
      constexpr int ch__ = (U) / 4, a__ = (U) % 4;                                             \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if ((U) + 1 < 8) GTXP_LOAD_A(((U) + 1) / 4, ((U) + 1) % 4, ((U) + 1) & 1)                \
      if (!DIRECT && ch__ == 0) GTXP_LOAD_B(1, a__)                                            \
      acc[a__][0] = GTXP_MFMA(al[(U) & 1], bh[ch__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXP_MFMA(al[(U) & 1], bh[ch__][1], acc[a__][1]);                          \
      acc[a__][0] = GTXP_MFMA(ah[(U) & 1], bl[ch__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXP_MFMA(ah[(U) & 1], bl[ch__][1], acc[a__][1]);                          \
      acc[a__][0] = GTXP_MFMA(ah[(U) & 1], bh[ch__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXP_MFMA(ah[(U) & 1], bh[ch__][1], acc[a__][1]);                          \
      _Pragma("unroll") for (int i__ = 0; i__ < 3; ++i__) {                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }

  for (int stage = 0; stage < nstages; ++stage) {
    __syncthreads();                               // the previous stage's fragment reads are done
    GTXP_COMMIT()
    __syncthreads();
    if (DIRECT) {                                  // this stage's pixels are the registers the previous stage prefetched
#pragma unroll
      for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {
        bh[s >> 1][s & 1] = __builtin_bit_cast(half8, pre_a[s]);
        bl[s >> 1][s & 1] = __builtin_bit_cast(half8, pre_b[s]);
      }
    }
    if (stage + 1 < nstages) GTXP_PREFETCH(stage + 1)
    if (!DIRECT) { GTXP_LOAD_B(0, 0) GTXP_LOAD_B(0, 1) GTXP_LOAD_B(0, 2) GTXP_LOAD_B(0, 3) }
    GTXP_LOAD_A(0, 0, 0)
    GTXP_UNIT(0) GTXP_UNIT(1) GTXP_UNIT(2) GTXP_UNIT(3)
    // an absent second chunk is zeros on both sides: its products add nothing (exact), the loop stays uniform
    GTXP_UNIT(4) GTXP_UNIT(5) GTXP_UNIT(6) GTXP_UNIT(7)
  }
#undef GTXP_UNIT
#undef GTXP_LOAD_A
#undef GTXP_LOAD_B

  // ---- epilogue: conv_k32_split.hip's (acc * 2^-shift -> activation (+ residual) -> split -> NHWC pair format) ----
  const float sc = P.acc_scale;
  const int cvalid = P.Cout - ct * BN;
  const bool plain = P.out_plain != 0;
  const int act = P.act;                          // 0 none, 1 SiLU, 2 ReLU
  const void* const res_p = P.res;
  float* const o_base = static_cast<float*>(P.out);
  bool sat = false;
  constexpr int PITCH = Tile::EPI_PITCH;
  __syncthreads();                                // every wave is done with the staging buffers
  char* stg = smem + wave * (32 * PITCH);
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int oy = oy0 + 2 * wave + m, ox = ox0 + col;
    const bool inside = oy < P.Ho && ox < P.Wo;
    const size_t pix = inside ? ((size_t)n * P.Ho + oy) * P.Wo + ox : 0;
    const float* __restrict__ res =
        (res_p && inside) ? static_cast<const float*>(res_p) + pix * P.res_cstride + P.res_coff + ct * BN : nullptr;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cl = 16 * a + 4 * kg;
      float2v v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        v[q] = float2v{acc[a][m][2 * q], acc[a][m][2 * q + 1]} * sc;
        if (act == 1) v[q] = silu2(v[q]); else if (act == 2) v[q] = relu2(v[q]);
      }
      if (res_p) {                                 // uniform; the swaps need every lane
        uint4 rc = make_uint4(0, 0, 0, 0);         // even kg: the group's hi chunk, odd kg: its lo chunk
        if (res && cl < cvalid) rc = *reinterpret_cast<const uint4*>(res + cl);
        const auto sx = __builtin_amdgcn_permlane16_swap(rc.x, rc.z, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(rc.y, rc.w, false, false);
        const unsigned hw[2] = {sx[0], sy[0]}, lw[2] = {sx[1], sy[1]};
        const half4 rh = *reinterpret_cast<const half4*>(hw), rl = *reinterpret_cast<const half4*>(lw);
#pragma unroll
        for (int q = 0; q < 2; ++q)
          v[q] += float2v{(float)rh[2 * q], (float)rh[2 * q + 1]} + float2v{(float)rl[2 * q], (float)rl[2 * q + 1]};
      }
      char* dst = stg + (16 * m + col) * PITCH + cl * 4;
      if (plain) {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
      } else {
        uint2 hi, lo;
        split2(v[0], hi.x, lo.x, sat);
        split2(v[1], hi.y, lo.y, sat);
        const auto sx = __builtin_amdgcn_permlane16_swap(hi.x, lo.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(hi.y, lo.y, false, false);
        *reinterpret_cast<uint4*>(dst) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
      }
    }
  }
  constexpr int LPP = BN / 4, PPI = 64 / LPP;     // 16 lanes of 16 B per pixel, 4 pixels per store instruction
#pragma unroll
  for (int it = 0; it < 32 / PPI; ++it) {
    const int p = it * PPI + lane / LPP, q = lane % LPP;
    const int py = oy0 + 2 * wave + (p >> 4), px = ox0 + (p & 15);
    const uint4 val = *reinterpret_cast<const uint4*>(stg + p * PITCH + q * 16);
    if (py < P.Ho && px < P.Wo && (q >> 1) * 8 < cvalid) {
      float* dst = o_base + (((size_t)n * P.Ho + py) * P.Wo + px) * P.out_cstride + P.out_coff + ct * BN + q * 4;
      *reinterpret_cast<uint4*>(dst) = val;
    }
  }
  if (P.sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(P.sat_flag, 1);
}

}  // namespace

void conv_k32p_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t stream) {
  GTX_CHECK(c.ks == 1 && c.stride == 1 && c.bn == K32PTile::BN && c.kc == K32PTile::KC && c.th == 8,
            "conv (K32 pointwise form): 1x1, 64-cout tiles, 32-channel chunks (ks=%d stride=%d bn=%d kc=%d)", c.ks, c.stride, c.bn, c.kc);
  for (int i = 0; i < g.count; ++i)
    GTX_CHECK(g.p[i].Cin % K32PTile::KC == 0 && g.p[i].post_w == nullptr && g.p[i].front_img == nullptr && g.p[i].c_split == 0,
              "conv (K32 pointwise form): Cin %d must be a multiple of 32 and the launch a plain 1x1 layer", g.p[i].Cin);
  const char* e = getenv("GTX_K32P");               // 1: the form that stages the pixels in LDS too (A/B; read per launch so that one process can compare)
  const bool direct = !(e && e[0] == '1');
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k32p_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, K32PTile::LDS_BYTES));
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k32p_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, K32PTile::LDS_DIRECT_BYTES));
  });
  if (direct) hipLaunchKernelGGL(conv_k32p_split_kernel<true>, dim3(g.grid_blocks), dim3(256), K32PTile::LDS_DIRECT_BYTES, stream, g);
  else hipLaunchKernelGGL(conv_k32p_split_kernel<false>, dim3(g.grid_blocks), dim3(256), K32PTile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
