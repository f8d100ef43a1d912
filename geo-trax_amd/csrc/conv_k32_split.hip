// 3x3 stride-1 split-f16x3 convolution on v_mfma_f32_16x16x32_f16 ("K32 form", ConvConfig::variant 5). gfx950 only.
//
// Same arithmetic contract as conv_igemm_split.hip (pair-format activations, every product as three fp16 MFMAs -- small terms
// first -- with fp32 accumulation, weights scaled by an exact power of two, same packed weight image with 32-channel chunks),
// the other matrix instruction: K = 32 per issue on a 16 x 16 tile. The chip holds a higher clock under that shape
// (MI355X_MICROARCH.md, "DVFS give-back": 1.12-1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP); a timing-only swap
// inside the 32x32x16 kernel priced it at 8 % on the matrix-bound layers (profiles/HISTORY.md section 8).
//
// K = 32 does not divide a tap's 16 staged channels, so a K chunk here is 32 input channels and a kernel ROW of the weights
// (3 taps x 64 couts x 128 B = 24.6 KB) is staged at a time beside the chunk's patch (10 x 18 pixels x 128 B = 23 KB):
// 47.6 KB per workgroup, three per CU like the 32x32x16 kernel. A workgroup = 8 x 16 output pixels x 64 couts; wave w owns
// tile rows 2w, 2w + 1 = two 16-pixel B blocks, and all four 16-cout A blocks: 8 accumulators of 4 registers.
// Fragment registers are the budget (3 waves per SIMD = 168 registers): the pixels' fragments (B) are held per tap and read one
// tap ahead, the weights' (A) per 16-cout block and read one block ahead -- 3 ds_read_b128 per 6 MFMAs, the 32x32x16 kernel's ratio.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <mutex>

#include "conv_igemm.hpp"

namespace gtx {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// Timing-only builds (`make k32probe`, wrong results): GTXK_PROBE bit 0 = no barriers inside a chunk (rows 1, 2), bit 1 = no weight
// commits (LDS stores) for rows 1, 2, bit 2 = no MFMAs, bit 3 = no weight loads from global for rows 1, 2, bit 5 = no weight fragment reads from LDS (the first tap's stay in registers),
// bit 6 = no pixel fragment reads from LDS after a chunk's first tap, bit 7 = the first chunk's global loads skipped (what a workgroup
// that had requested them under the previous tile's epilogue would see), bit 8 = no epilogue (one store per lane)
#ifndef GTXK_PROBE
#define GTXK_PROBE 0
#endif
#if GTXK_PROBE & 4
#define GTXK_MFMA(a, b, c) (c + floatx4{(float)(a)[0], (float)(b)[0], 0.f, 0.f})
#else
#define GTXK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
#define GTXK_SYNC_IN() { if (!(GTXK_PROBE & 1)) __syncthreads(); }

struct K32Tile {
  static constexpr int TH = 8, TW = 16, BN = 64, KC = 32, CPR = 4, NCH = 8, RB = 128;
  static constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
  static constexpr int PATCH_UNITS = NPIX * CPR;                  // one unit = 8 channels of one pixel (hi chunk, lo chunk)
  static constexpr int PATCH_SLOTS = (PATCH_UNITS + 255) / 256;   // 3
  static constexpr int PATCH_BYTES = NPIX * RB;
  static constexpr int WROW_CHUNKS = 3 * BN * NCH;                // 16-byte chunks of one kernel row of weights: 1536
  static constexpr int W_SLOTS = WROW_CHUNKS / 256;               // 6
  static constexpr int WROW_BYTES = WROW_CHUNKS * 16;
  static constexpr int STAGE_BYTES = PATCH_BYTES + WROW_BYTES;
  static constexpr int EPI_PITCH = BN * 4 + 16;
  static constexpr int EPI_BYTES = 4 * 32 * EPI_PITCH;
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static __host__ __device__ constexpr int swz(int row) { return (row >> 1) & 7; }   // conv_igemm_split.hip's swizzle for 128-byte rows
};
static_assert(3 * K32Tile::LDS_BYTES <= 160 * 1024, "three workgroups per CU");

// conv_igemm_split.hip's epilogue arithmetic (same operations in the same order)
__device__ __forceinline__ float2v relu2(const float2v v) { return float2v{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)}; }   // ConvProblem::act == 2 (RT-DETR's HGNetv2 blocks)
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

__global__ __attribute__((amdgpu_flat_work_group_size(1, 256), amdgpu_waves_per_eu(3)))
void conv_k32_split_kernel(const ConvGroup g) {
  using Tile = K32Tile;
  constexpr int PW = Tile::PW, RB = Tile::RB, BN = Tile::BN, CPR = Tile::CPR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + Tile::PATCH_BYTES;

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
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;

  const float* __restrict__ in = static_cast<const float*>(P.in);
  const int nchunks = P.Cin / Tile::KC;

  int goff[Tile::PATCH_SLOTS];                    // element offset of the unit (activation buffers are < 2^31 elements), -1 = zero fill
  int loff[Tile::PATCH_SLOTS];                    // LDS byte offset of the unit's hi chunk (-1 = unused slot); its lo chunk: ^ 64
#pragma unroll
  for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {
    const int qid = tid + 256 * s;
    const int p = qid / CPR, c = qid % CPR;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool used = qid < Tile::PATCH_UNITS;
    const bool inb = used && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    goff[s] = inb ? ((n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff + c * 8 : -1;
    loff[s] = used ? p * RB + ((c ^ Tile::swz(p)) << 4) : -1;
  }
  // packed image: [cout tile][chunk][tap = 3 ky + kx][n][8 swizzled 16-byte chunks] (pack_conv_weights_split, kc = 32): a kernel
  // row of a chunk is 1536 contiguous uint4
  const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>(P.wpack) + (size_t)ct * nchunks * (3 * Tile::WROW_CHUNKS) + tid;

  const int col = lane & 15, kg = lane >> 4;
  const int p0 = (2 * wave) * PW + col;            // patch pixel of (tile row 2 wave, column col) at tap (0, 0)

  uint4 pre_a[Tile::PATCH_SLOTS], pre_b[Tile::PATCH_SLOTS];
  uint4 pw0, pw1, pw2, pw3, pw4, pw5;               // the next kernel row of weights (W_SLOTS = 6)
  static_assert(Tile::W_SLOTS == 6, "six weight slots");
#define GTXK_PREFETCH_PATCH(CHUNK)                                                           \
  {                                                                                          \
    const int c0__ = (CHUNK) * Tile::KC;                                                     \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      uint4 va__ = make_uint4(0, 0, 0, 0), vb__ = make_uint4(0, 0, 0, 0);                    \
      if (goff[s] >= 0) {                                                                    \
        const uint4* src__ = reinterpret_cast<const uint4*>(in + goff[s] + c0__);            \
        va__ = src__[0];                                                                     \
        vb__ = src__[1];                                                                     \
      }                                                                                      \
      pre_a[s] = va__;                                                                       \
      pre_b[s] = vb__;                                                                       \
    }                                                                                        \
  }
#define GTXK_PREFETCH_W(ROWIDX)                      /* ROWIDX = 3 chunk + kernel row */      \
  {                                                                                          \
    const uint4* w__ = wsrc + (size_t)(ROWIDX) * Tile::WROW_CHUNKS;                          \
    pw0 = w__[0]; pw1 = w__[256]; pw2 = w__[512]; pw3 = w__[768]; pw4 = w__[1024]; pw5 = w__[1280]; \
  }
#define GTXK_COMMIT_PATCH()                                                                  \
  {                                                                                          \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      if (loff[s] >= 0) {                                                                    \
        *reinterpret_cast<uint4*>(lds_patch + loff[s]) = pre_a[s];                           \
        *reinterpret_cast<uint4*>(lds_patch + (loff[s] ^ (CPR << 4))) = pre_b[s];                           \
      }                                                                                      \
    }                                                                                        \
  }
#define GTXK_COMMIT_W()                                                                      \
  {                                                                                          \
    uint4* d__ = reinterpret_cast<uint4*>(lds_w) + tid;                                      \
    d__[0] = pw0; d__[256] = pw1; d__[512] = pw2; d__[768] = pw3; d__[1024] = pw4; d__[1280] = pw5; \
  }

  if (GTXK_PROBE & 128) {                          // timing-only: the first chunk's loads cost nothing
#pragma unroll
    for (int s = 0; s < Tile::PATCH_SLOTS; ++s) pre_a[s] = pre_b[s] = make_uint4(tid, s, 0, 0);
    pw0 = pw1 = pw2 = pw3 = pw4 = pw5 = make_uint4(tid, 1, 2, 3);
  } else {
    GTXK_PREFETCH_PATCH(0)
    GTXK_PREFETCH_W(0)
  }

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

  half8 bh[2][2], bl[2][2];                      // [slot][pixel block m]
  half8 ah[2], al[2];                            // [slot]
  // pixels' fragments of tap (KY, KX) -> slot; piece Q = 0..3 is one 16-byte read (m = Q >> 1, hi / lo = Q & 1)
#define GTXK_LOAD_B(KY, KX, SLOT, Q)                                                           \
    {                                                                                          \
      const int p__ = p0 + (((Q) >> 1) + (KY)) * PW + (KX);                                    \
      const char* pr__ = lds_patch + p__ * RB;                                                 \
      if (((Q) & 1) == 0) bh[SLOT][(Q) >> 1] = *reinterpret_cast<const half8*>(pr__ + ((kg ^ Tile::swz(p__)) << 4)); \
      else bl[SLOT][(Q) >> 1] = *reinterpret_cast<const half8*>(pr__ + (((CPR + kg) ^ Tile::swz(p__)) << 4)); \
    }
  // weights' fragments of (tap column KX, cout block A) -> slot
#define GTXK_LOAD_A(KX, A, SLOT)                                                               \
    {                                                                                          \
      const int nrow__ = 16 * (A) + col;                                                       \
      const char* wr__ = lds_w + ((KX) * BN + nrow__) * RB;                                    \
      ah[SLOT] = *reinterpret_cast<const half8*>(wr__ + ((kg ^ Tile::swz(nrow__)) << 4));     \
      al[SLOT] = *reinterpret_cast<const half8*>(wr__ + (((CPR + kg) ^ Tile::swz(nrow__)) << 4)); \
    }
  // One kernel row = 12 units (tap column kx, cout block a) of 6 MFMAs; unit u reads the weights of unit u + 1 and one piece of
  // the next tap's pixels (the next row's first tap too: the patch stays through the chunk). BSLOT0 = slot of the row's first tap.
#define GTXK_UNIT(KY, U, LAST_ROW)                                                             \
    {                                                                                          \
      constexpr int kx__ = (U) / 4, a__ = (U) % 4;                                             \
      constexpr int bs__ = ((KY) * 3 + kx__) & 1;                                              \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if ((U) + 1 < 12 && !(GTXK_PROBE & 32)) GTXK_LOAD_A(((U) + 1) / 4, ((U) + 1) % 4, ((U) + 1) & 1) \
      if (GTXK_PROBE & 64) {}                                                                  \
      else if (kx__ < 2) GTXK_LOAD_B(KY, kx__ + 1, bs__ ^ 1, a__)                              \
      else if (!(LAST_ROW)) GTXK_LOAD_B((KY) + 1, 0, bs__ ^ 1, a__)                            \
      acc[a__][0] = GTXK_MFMA(al[(U) & 1], bh[bs__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXK_MFMA(al[(U) & 1], bh[bs__][1], acc[a__][1]);                          \
      acc[a__][0] = GTXK_MFMA(ah[(U) & 1], bl[bs__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXK_MFMA(ah[(U) & 1], bl[bs__][1], acc[a__][1]);                          \
      acc[a__][0] = GTXK_MFMA(ah[(U) & 1], bh[bs__][0], acc[a__][0]);                          \
      acc[a__][1] = GTXK_MFMA(ah[(U) & 1], bh[bs__][1], acc[a__][1]);                          \
      _Pragma("unroll") for (int i__ = 0; i__ < 3; ++i__) {                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }
#define GTXK_ROW(KY, LAST_ROW)                                                                 \
    GTXK_LOAD_A(0, 0, 0) if (GTXK_PROBE & 32) GTXK_LOAD_A(0, 1, 1) if (GTXK_PROBE & 64) { bh[1][0] = bh[0][0]; bh[1][1] = bh[0][1]; bl[1][0] = bl[0][0]; bl[1][1] = bl[0][1]; }                                                                      \
    GTXK_UNIT(KY, 0, LAST_ROW) GTXK_UNIT(KY, 1, LAST_ROW) GTXK_UNIT(KY, 2, LAST_ROW) GTXK_UNIT(KY, 3, LAST_ROW)   \
    GTXK_UNIT(KY, 4, LAST_ROW) GTXK_UNIT(KY, 5, LAST_ROW) GTXK_UNIT(KY, 6, LAST_ROW) GTXK_UNIT(KY, 7, LAST_ROW)   \
    GTXK_UNIT(KY, 8, LAST_ROW) GTXK_UNIT(KY, 9, LAST_ROW) GTXK_UNIT(KY, 10, LAST_ROW) GTXK_UNIT(KY, 11, LAST_ROW)

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // ---- kernel row 0: new patch + the row's weights ----
    __syncthreads();                               // the previous chunk's fragment reads are done
    GTXK_COMMIT_PATCH()
    GTXK_COMMIT_W()
    __syncthreads();
    if (!(GTXK_PROBE & 8)) GTXK_PREFETCH_W(3 * chunk + 1)
    GTXK_LOAD_B(0, 0, 0, 0) GTXK_LOAD_B(0, 0, 0, 1) GTXK_LOAD_B(0, 0, 0, 2) GTXK_LOAD_B(0, 0, 0, 3)
    GTXK_ROW(0, false)
    // ---- kernel row 1 ----
    GTXK_SYNC_IN()
    if (!(GTXK_PROBE & 2)) GTXK_COMMIT_W()
    GTXK_SYNC_IN()
    if (!(GTXK_PROBE & 8)) GTXK_PREFETCH_W(3 * chunk + 2)
    GTXK_ROW(1, false)
    // ---- kernel row 2: the next chunk's patch and first row are requested here ----
    GTXK_SYNC_IN()
    if (!(GTXK_PROBE & 2)) GTXK_COMMIT_W()
    GTXK_SYNC_IN()
    if (chunk + 1 < nchunks) {
      GTXK_PREFETCH_PATCH(chunk + 1)
      GTXK_PREFETCH_W(3 * chunk + 3)
    }
    GTXK_ROW(2, true)
  }
#undef GTXK_ROW
#undef GTXK_UNIT
#undef GTXK_LOAD_A
#undef GTXK_LOAD_B

  // ---- epilogue: acc * 2^-shift -> SiLU (+ residual) -> split -> NHWC pair format ----
  // Lane (col, kg) of block (a, m) holds couts 16 a + 4 kg + 0..3 of pixel (row 2 wave + m, col): lanes kg = 2 q and 2 q + 1 hold
  // the two halves of the 8-channel group 2 a + q, and two v_permlane16_swap turn them into the group's 16-byte hi chunk (even kg)
  // and lo chunk (odd kg) -- byte 64 a + 16 kg of the pixel's 256-byte run. Staged per wave in LDS, stored as whole runs.
  if (GTXK_PROBE & 256) {
    floatx4 t = acc[0][0];
#pragma unroll
    for (int a = 0; a < 4; ++a) { t += acc[a][0]; t += acc[a][1]; }
    const int oy = oy0 + 2 * wave, ox = ox0 + col;
    if (oy < P.Ho && ox < P.Wo) *reinterpret_cast<floatx4*>(static_cast<float*>(P.out) + (((size_t)n * P.Ho + oy) * P.Wo + ox) * P.out_cstride + P.out_coff + ct * BN + 4 * kg) = t;
    return;
  }
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

void conv_k32_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t stream) {
  GTX_CHECK(c.ks == 3 && c.stride == 1 && c.bn == K32Tile::BN && c.kc == K32Tile::KC && c.th == 8,
            "conv (K32 form): 3x3 stride 1, 64-cout tiles, 32-channel chunks (ks=%d stride=%d bn=%d kc=%d)", c.ks, c.stride, c.bn, c.kc);
  for (int i = 0; i < g.count; ++i)
    GTX_CHECK(g.p[i].Cin % K32Tile::KC == 0 && g.p[i].post_w == nullptr && g.p[i].front_img == nullptr && g.p[i].c_split == 0,
              "conv (K32 form): Cin %d must be a multiple of 32 and the launch a plain 3x3 layer", g.p[i].Cin);
  auto kern = conv_k32_split_kernel;
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, K32Tile::LDS_BYTES));
  });
  hipLaunchKernelGGL(kern, dim3(g.grid_blocks), dim3(256), K32Tile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
