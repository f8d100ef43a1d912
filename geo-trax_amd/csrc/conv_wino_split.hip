// Winograd F(2x2, 3x3) form of the fp32-grade ("split-f16x3") 3x3 stride-1 convolution. gfx950 only.
//
// Same arithmetic contract as conv_igemm_split.hip (activations in pair format, every product as three fp16 MFMAs with fp32
// accumulation, weights scaled by an exact power of two), 2.25 x fewer matrix instructions: a 4x4 input tile d and the 3x3
// weights g of a (cin, cout) pair give the 2x2 output tile
//     Y = At [ (G g Gt) (.) (Bt d B) ] A            ((.) = element by element, summed over cin)
// so the sixteen positions of the transformed tile are sixteen independent GEMMs [cout x cin] x [cin x tiles].
//   * U = G g Gt is made on the host in float64, scaled and split into hi + lo fp16 halves (pack_conv_weights_wino);
//   * V = Bt d B is made by the workgroup with vector instructions from the pair-format patch it staged in LDS: d = hi + lo is
//     put back together in fp32 (v_fma_mix_f32 reads the fp16 halves directly), transformed (coefficients +-1: no rounding
//     beyond fp32's), and split again into the two fp16 numbers the matrix pipe consumes -- 22 significand bits of V;
//   * the products run as v_mfma_f32_32x32x16_f16 D[cout 32][tile 32] per position, three per (position, cout block);
//   * Y = At M A: the four positions of a transform row live in one wave (column stage in registers), the four rows in four
//     waves (row stage through LDS, each wave finishing a quarter of the channels), then bias, SiLU, residual, split, store.
//
// Work decomposition: one 512-thread workgroup (8 waves) = 8 x 16 output pixels (4 x 8 Winograd tiles = the 32 columns of
// the MFMA) x 64 output channels. Wave w works on transform row r = w >> 1 and cout block j = w & 1: per 16-channel K chunk
// it issues 4 positions x 3 = 12 MFMAs whose A operands (U) come STRAIGHT FROM GLOBAL MEMORY into registers -- a position's
// weights are used by exactly one wave per cout block, the packed image is in fragment order, there is nothing to share
// through LDS -- and whose B operands (V) come from LDS. For the transform the same wave pair (2r, 2r + 1) produces row r of V:
// a thread takes one tile, four channels and the row's four positions (two patch rows x four columns in, sixteen values out).
// Per chunk the loop overlaps, in one basic block between two barriers: the MFMAs of chunk i, the transform of chunk i + 1
// (raw patch buffer (i + 1) & 1 -> V buffer (i + 1) & 1) and the global loads of chunk i + 2's patch and chunk i + 1's weights.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <mutex>

#include "conv_igemm.hpp"

namespace gtx {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#define GTXW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// Timing-only builds (`make winoprobe`, wrong results): GTXW_PROBE bit 0 = no transform arithmetic (the values read are
// written back unchanged), bit 1 = no MFMAs, bit 2 = no transform at all (no LDS reads / writes of it either), bit 3 = no weight
// loads inside the loop, bit 4 = no patch loads / commits inside the loop, bit 5 = no epilogue (one store per lane), bit 6 = no
// barrier inside the loop
#ifndef GTXW_PROBE
#define GTXW_PROBE 0
#endif

struct WinoTile {
  static constexpr int TH = 8, TW = 16, BN = 64, KC = 16;
  static constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;     // 10 x 18 input pixels
  static constexpr int RAW_UNITS = NPIX * 2;                         // (pixel, 8-channel unit): 360 of the 512 threads load one
  static constexpr int RAW_BYTES = (NPIX + 1) * 64;                  // a pixel's 16 channels: four quads of [4 hi halves | 4 lo halves]; + one row nobody reads
                                                                     // (where the 152 threads without a unit put their zeros: no branch in the loop)
  static constexpr int NT = 32;                                      // Winograd tiles = MFMA columns
  static constexpr int V_BYTES = 16 * NT * 64;                       // [position][tile][hi0 hi1 lo0 lo1]: 32 KB
  static constexpr int LOOP_BYTES = 2 * RAW_BYTES + 2 * V_BYTES;     // both double-buffered: 88 576 B
  static constexpr int X_BYTES = 2 * 4 * 4 * 2 * 64 * 16;            // row-stage exchange [j][dst r][src r][b][lane] float4: 64 KB
  static constexpr int LDS_BYTES = LOOP_BYTES > X_BYTES ? LOOP_BYTES : X_BYTES;
  // patch pixel (py, px) -> its 64-byte row: the even columns of a patch row, then the odd ones, so that the tiles tx and
  // tx + 1 (columns two apart) read neighbouring rows: eight lanes of a ds_read_b128 cover 128 contiguous bytes
  static __host__ __device__ constexpr int raw_row(int py, int px) { return py * PW + (px & 1) * (PW / 2) + (px >> 1); }
  // 16-byte chunk swizzle of a V row by its tile: spreads the half-row writes of the transform (32 B per tile) and the
  // 16-byte fragment reads of the MFMAs over the banks
  static __host__ __device__ constexpr int v_swz(int t) { return (2 * ((t >> 1) & 1)) ^ ((t >> 2) & 1); }
};

__device__ __forceinline__ float2v silu2(const float2v v) {
  const float2v t = v * -1.44269504088896341f;
  const float2v d = float2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.f;
  return v * float2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// 2 fp32 values -> hi halves, lo halves (conv_igemm_split.hip's split2: clamp to +-65504, flag what was clamped, NaN included)
__device__ __forceinline__ void split2(const float2v v, unsigned& hi, unsigned& lo, bool& sat) {
  const float2v x = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
  sat |= x.x != v.x || x.y != v.y;
  const half2v h = __builtin_convertvector(x, half2v);
  const half2v l = __builtin_convertvector(x - __builtin_convertvector(h, float2v), half2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// ... without the clamp, for the transformed inputs: a V beyond fp16's range becomes inf, its lo half -inf or NaN, the
// products NaN, and the epilogue's split2 flags the NaN (ConvProblem::sat_flag -> the detector falls back to exact fp32)
__device__ __forceinline__ void split2_raw(const float2v x, unsigned& hi, unsigned& lo) {
  const half2v h = __builtin_convertvector(x, half2v);
  const half2v l = __builtin_convertvector(x - __builtin_convertvector(h, float2v), half2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// d = hi + lo is put back together, two pixels are combined and x - hi is taken for the split by fp32 fmas whose sources are
// fp16 halves: the compiler selects v_fma_mix_f32 for them (this file is built with -fno-slp-vectorize: paired up, the same
// expressions become v_cvt_f32_f16 x 4 + v_pk_fma_f32, twice the instructions).
// (a_hi + a_lo) + sgn * (b_hi + b_lo) for the two values packed in the registers, small terms first. `one` is 1.0 in a register:
// with the literal the compiler folds the fma into an add and converts the half first
__device__ __forceinline__ float2v combine2(unsigned ah, unsigned al, unsigned bh, unsigned bl, float sgn, float one) {
  const half2v ah2 = __builtin_bit_cast(half2v, ah), al2 = __builtin_bit_cast(half2v, al);
  const half2v bh2 = __builtin_bit_cast(half2v, bh), bl2 = __builtin_bit_cast(half2v, bl);
  float2v x;
  x.x = __builtin_fmaf((float)bh2[0], sgn, __builtin_fmaf((float)ah2[0], one, __builtin_fmaf((float)bl2[0], sgn, (float)al2[0])));
  x.y = __builtin_fmaf((float)bh2[1], sgn, __builtin_fmaf((float)ah2[1], one, __builtin_fmaf((float)bl2[1], sgn, (float)al2[1])));
  return x;
}
// x -> hi = fp16(x), lo = fp16(x - hi), no clamp (see split2_raw)
__device__ __forceinline__ void split2_mix(const float2v x, unsigned& hi, unsigned& lo, float mone) {
  const half2v h = __builtin_convertvector(x, half2v);
  hi = __builtin_bit_cast(unsigned, h);
  const float2v d = {__builtin_fmaf((float)h[0], mone, x.x), __builtin_fmaf((float)h[1], mone, x.y)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, half2v));
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wino_split_kernel(const ConvGroup g) {
  using T = WinoTile;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const lds_raw = smem;                        // [2][RAW_BYTES]
  char* const lds_v = smem + 2 * T::RAW_BYTES;       // [2][V_BYTES]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = wave >> 1, j = wave & 1;             // transform row of this wave, its 32-cout block

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
  const int tx0 = pt % P.tiles_x;
  const int t2 = pt / P.tiles_x;
  const int ty0 = t2 % P.tiles_y + P.ty_first;
  const int n = t2 / P.tiles_y;
  const int oy0 = ty0 * T::TH, ox0 = tx0 * T::TW;
  const int nchunks = P.Cin / T::KC;
  const float* __restrict__ in = static_cast<const float*>(P.in);

  // ---- raw patch: one (pixel, 8-channel unit) per thread, 360 of 512. Buffer loads: a pixel outside the image (the layer's
  // zero padding) and the threads without a unit read at an offset past the tensor's end, which returns zeros -- no branch ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((size_t)P.N * P.H * P.W * P.in_cstride * 4), 0x00020000);
  unsigned gvoff = 0x80000000u;                       // byte offset of the unit's first chunk; this one is past any tensor
  int rdst = T::NPIX * 64;                            // the dummy row
  if (tid < T::RAW_UNITS) {
    const int pidx = tid >> 1, c = tid & 1;
    const int py = pidx / T::PW, px = pidx - py * T::PW;
    const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
    if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) gvoff = (unsigned)((((size_t)(n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff + c * 8) * 4);
    rdst = T::raw_row(py, px) * 64 + c * 32;
  }
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  // Two register sets: the unit's hi chunk and lo chunk of chunk i + 2 (set i & 1, committed at the end of iteration i) and of
  // chunk i + 3 (the other set) are on their way in while iteration i runs -- a load has two iterations to arrive. With one
  // workgroup per CU there is no other workgroup to hide an exposed HBM round trip (0.7 us per chunk with a single set).
  uint4v ra0_, rb0_, ra1_, rb1_;
  const int last_chunk_off = (nchunks - 1) * (T::KC * 4);
#define GTXW_RAW_LOAD(S, CHUNK)                      /* chunks past the last one: the last one again (never used) */ \
  {                                                                                          \
    const int so__ = min((CHUNK) * (T::KC * 4), last_chunk_off);                             \
    ra##S##_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gvoff, so__, 0);               \
    rb##S##_ = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gvoff, so__ + 16, 0);          \
  }
  // quad q of the unit = [4 hi halves | 4 lo halves]: 16 bytes the transform reads with one ds_read_b128
#define GTXW_RAW_COMMIT(S, BUF)                                                              \
  {                                                                                          \
    char* d__ = lds_raw + (BUF) * T::RAW_BYTES + rdst;                                       \
    *reinterpret_cast<uint4*>(d__) = make_uint4(ra##S##_.x, ra##S##_.y, rb##S##_.x, rb##S##_.y); \
    *reinterpret_cast<uint4*>(d__ + 16) = make_uint4(ra##S##_.z, ra##S##_.w, rb##S##_.z, rb##S##_.w); \
  }

  // ---- weights: this wave's (r, j) fragments of a chunk, 8 x 16 bytes per lane: position c hi, position c lo ----
  const uint4* __restrict__ usrc = reinterpret_cast<const uint4*>(P.wpack) + ((((size_t)ct * nchunks) * 4 + r) * 2 + j) * (8 * 64) + lane;
  constexpr size_t U_CHUNK = 4 * 2 * 8 * 64;          // uint4 per K chunk (all rows, both cout blocks)
  // Two register sets: chunk i + 1's eight fragments are requested TOGETHER at the top of iteration i. (Reloading each
  // position's pair behind its MFMAs into one set spread the requests over the iteration: a wave's vector and matrix
  // instructions then queue behind its own loads whenever the CU's vector memory path -- 64 B/clk, 75 KB per chunk: the
  // bound of this loop -- is backed up, and the compute time ADDED to the memory time instead of hiding under it.)
  uint4 u0[8], u1[8];
#define GTXW_U_LOAD(S, CHUNK) { _Pragma("unroll") for (int e__ = 0; e__ < 8; ++e__) u##S[e__] = usrc[(size_t)(CHUNK) * U_CHUNK + e__ * 64]; }

  // ---- transform job of this thread: tile tt, channel quad q, row r (wave-uniform) ----
  const int tidx = (wave & 1) * 64 + lane;
  const int tt = tidx >> 2, q = tidx & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  // Bt d, row r:  r0 = d0 - d2,  r1 = d1 + d2,  r2 = d2 - d1,  r3 = d1 - d3
  const int row_a = r == 0 ? 0 : (r == 2 ? 2 : 1);
  const int row_b = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
  const float sgn = r == 1 ? 1.f : -1.f;
  const float one = P.acc_scale * __builtin_amdgcn_rcpf(P.acc_scale);   // 1.0 (acc_scale is a power of two) the compiler cannot see through (combine2)
  const int t_base = ((2 * tty) * T::PW + ttx) * 64 + q * 16;
  const int ta_off = t_base + row_a * T::PW * 64, tb_off = t_base + row_b * T::PW * 64;
  const int vsw_t = T::v_swz(tt);
  const int vw_hi = ((4 * r) * T::NT + tt) * 64 + (((q >> 1) ^ vsw_t) << 4) + (q & 1) * 8;
  const int vw_lo = ((4 * r) * T::NT + tt) * 64 + (((2 + (q >> 1)) ^ vsw_t) << 4) + (q & 1) * 8;
#define GTXW_T_READ(RBUF)                              /* the job's 2 x 4 input pixels */   \
    uint4 A__[4], B__[4];                                                                    \
    {                                                                                        \
      const char* pa__ = lds_raw + (RBUF) * T::RAW_BYTES + ta_off;                           \
      const char* pb__ = lds_raw + (RBUF) * T::RAW_BYTES + tb_off;                           \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                        \
        const int co__ = ((c & 1) * (T::PW / 2) + (c >> 1)) * 64;                            \
        A__[c] = *reinterpret_cast<const uint4*>(pa__ + co__);                               \
        B__[c] = *reinterpret_cast<const uint4*>(pb__ + co__);                               \
      }                                                                                      \
    }
// The arithmetic of the transform in four parts of ~24 vector instructions, one behind each position's three MFMAs:
// X = rows combined (per patch column), V = columns combined ((Bt d) B: X0 - X2, X1 + X2, X2 - X1, X1 - X3), split, written.
#if GTXW_PROBE & 1
#define GTXW_T_X(C) X__[C][0] = float2v{__builtin_bit_cast(float, A__[C].x), __builtin_bit_cast(float, B__[C].x)}; X__[C][1] = float2v{__builtin_bit_cast(float, A__[C].y), __builtin_bit_cast(float, B__[C].y)};
#define GTXW_T_PUT(C, V0, V1) { *reinterpret_cast<uint2*>(vwh__ + (C) * (T::NT * 64)) = make_uint2(__builtin_bit_cast(unsigned, X__[C][0].x), __builtin_bit_cast(unsigned, X__[C][0].y)); *reinterpret_cast<uint2*>(vwl__ + (C) * (T::NT * 64)) = make_uint2(__builtin_bit_cast(unsigned, X__[C][1].x), __builtin_bit_cast(unsigned, X__[C][1].y)); }
#else
#define GTXW_T_X(C)                                                                          \
    X__[C][0] = combine2(A__[C].x, A__[C].z, B__[C].x, B__[C].z, sgn, one);   /* (a_hi + a_lo) +- (b_hi + b_lo), channels 0, 1 */ \
    X__[C][1] = combine2(A__[C].y, A__[C].w, B__[C].y, B__[C].w, sgn, one);   /* channels 2, 3 */
#define GTXW_T_PUT(C, V0, V1)                                                                \
    {                                                                                        \
      uint2 hi__, lo__;                                                                      \
      split2_mix(V0, hi__.x, lo__.x, -one);                                                  \
      split2_mix(V1, hi__.y, lo__.y, -one);                                                  \
      *reinterpret_cast<uint2*>(vwh__ + (C) * (T::NT * 64)) = hi__;                          \
      *reinterpret_cast<uint2*>(vwl__ + (C) * (T::NT * 64)) = lo__;                          \
    }
#endif
#define GTXW_T_PART(PART)                                                                    \
    if (PART == 0) { GTXW_T_X(0) GTXW_T_X(1) }                                               \
    if (PART == 1) { GTXW_T_X(2) GTXW_T_PUT(0, X__[0][0] - X__[2][0], X__[0][1] - X__[2][1]) } \
    if (PART == 2) { GTXW_T_X(3) GTXW_T_PUT(1, X__[1][0] + X__[2][0], X__[1][1] + X__[2][1]) } \
    if (PART == 3) { GTXW_T_PUT(2, X__[2][0] - X__[1][0], X__[2][1] - X__[1][1]) GTXW_T_PUT(3, X__[1][0] - X__[3][0], X__[1][1] - X__[3][1]) }
#define GTXW_T_MATH(VBUF)                                                                    \
  {                                                                                          \
    float2v X__[4][2];                                                                       \
    char* vwh__ = lds_v + (VBUF) * T::V_BYTES + vw_hi;                                       \
    char* vwl__ = lds_v + (VBUF) * T::V_BYTES + vw_lo;                                       \
    GTXW_T_PART(0) GTXW_T_PART(1) GTXW_T_PART(2) GTXW_T_PART(3)                              \
  }

  // ---- MFMA operand addresses: column = tile (lane & 31), k half h ----
  const int mt = lane & 31, h = lane >> 5;
  const int vsw_m = T::v_swz(mt);
  const int vr_hi = ((4 * r) * T::NT + mt) * 64 + ((h ^ vsw_m) << 4);
  const int vr_lo = ((4 * r) * T::NT + mt) * 64 + (((2 + h) ^ vsw_m) << 4);

  floatx16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  // ---- prologue: chunks 0 .. 3 requested at once; chunk 0 staged and transformed, chunk 1 staged ----
  GTXW_RAW_LOAD(0, 0)
  GTXW_RAW_LOAD(1, 1)
  GTXW_U_LOAD(0, 0)
  if (GTXW_PROBE & 8) GTXW_U_LOAD(1, 0)
  const float4 bias4 = P.bias ? *reinterpret_cast<const float4*>(P.bias + ct * T::BN + 32 * j + 8 * r + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  GTXW_RAW_COMMIT(0, 0)
  GTXW_RAW_LOAD(0, 2)
  __syncthreads();
  {
    GTXW_T_READ(0)
    GTXW_T_MATH(0)
  }
  GTXW_RAW_COMMIT(1, 1)
  GTXW_RAW_LOAD(1, 3)
  __syncthreads();

  // One chunk of the loop, between two barriers, as ONE basic block without branches: the MFMAs of chunk i (V buffer i & 1),
  // behind each position's three MFMAs the reload of that position's weights for chunk i + 1 into the same registers (three
  // quarters of a chunk ahead of their use: an L2 hit) and a quarter of the transform of chunk i + 1 (raw buffer (i + 1) & 1 ->
  // V buffer (i + 1) & 1); at the end chunk i + 2's patch goes from its register set into raw buffer i & 1 (read by the
  // transform of chunk i, one barrier ago) and the set is sent for chunk i + 4. Chunks past the end are the last chunk again,
  // loaded and staged and never read. The last chunk only has its MFMAs.
#define GTXW_CHUNK(I, S, FULL)                       /* S = I & 1 at compile time */          \
  {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (FULL && !(GTXW_PROBE & 8)) { if (S == 0) GTXW_U_LOAD(1, (I) + 1) else GTXW_U_LOAD(0, (I) + 1) } \
    half8 bh__[4], bl__[4];                                                                  \
    {                                                                                        \
      const char* vh__ = lds_v + S * T::V_BYTES + vr_hi;                                     \
      const char* vl__ = lds_v + S * T::V_BYTES + vr_lo;                                     \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                        \
        bh__[c] = *reinterpret_cast<const half8*>(vh__ + c * (T::NT * 64));                  \
        bl__[c] = *reinterpret_cast<const half8*>(vl__ + c * (T::NT * 64));                  \
      }                                                                                      \
    }                                                                                        \
    GTXW_T_READ(S ^ 1)                            /* unused (and dropped) in the last chunk */ \
    float2v X__[4][2];                                                                       \
    char* vwh__ = lds_v + (S ^ 1) * T::V_BYTES + vw_hi;                                      \
    char* vwl__ = lds_v + (S ^ 1) * T::V_BYTES + vw_lo;                                      \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                          \
      const half8 uh__ = *reinterpret_cast<const half8*>(S == 0 ? &u0[2 * c] : &u1[2 * c]);  \
      const half8 ul__ = *reinterpret_cast<const half8*>(S == 0 ? &u0[2 * c + 1] : &u1[2 * c + 1]); \
      if (!(GTXW_PROBE & 2)) {                                                               \
      acc[c] = GTXW_MFMA(ul__, bh__[c], acc[c]);                                             \
      acc[c] = GTXW_MFMA(uh__, bl__[c], acc[c]);                                             \
      acc[c] = GTXW_MFMA(uh__, bh__[c], acc[c]);                                             \
      } else { acc[c][0] += (float)ul__[0] + (float)bh__[c][0] + (float)bl__[c][0] + (float)uh__[0]; } \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (FULL && !(GTXW_PROBE & 4)) { GTXW_T_PART(c) }                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    if (FULL && !(GTXW_PROBE & 16)) {                                                        \
      GTXW_RAW_COMMIT(S, S)                                                                  \
      GTXW_RAW_LOAD(S, (I) + 4)                                                              \
    }                                                                                        \
    if (!(GTXW_PROBE & 64)) __syncthreads();                                                 \
  }
  int i = 0;
  for (; i + 2 < nchunks; i += 2) {
    GTXW_CHUNK(i, 0, true)
    GTXW_CHUNK(i + 1, 1, true)
  }
  if (i + 1 < nchunks) {
    GTXW_CHUNK(i, 0, true)
    GTXW_CHUNK(i + 1, 1, false)
  } else {
    GTXW_CHUNK(i, 0, false)
  }
#undef GTXW_CHUNK
#undef GTXW_T_MATH
#undef GTXW_T_PART
#undef GTXW_T_PUT
#undef GTXW_T_X
#undef GTXW_T_READ
#undef GTXW_U_LOAD
#undef GTXW_RAW_COMMIT
#undef GTXW_RAW_LOAD

#if GTXW_PROBE & 32
  {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int k = 0; k < 16; ++k) v += acc[c][k];
    static_cast<float*>(P.out)[((size_t)blockIdx.x * 512 + tid) % 4096] = v + bias4.x;
    return;
  }
#endif
  // ---- output transform. Column stage in registers: T[r][0] = M0 + M1 + M2, T[r][1] = M1 - M2 - M3 ----
  // (the barrier that closed the loop also ends every wave's use of the loop's LDS buffers)
  {
    float4* xw = reinterpret_cast<float4*>(smem) + (((size_t)j * 4) * 4 + r) * 2 * 64 + lane;     // [j][dst][src = r][b][lane]
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
      float4 t0, t1;
      t0.x = acc[0][4 * rd + 0] + acc[1][4 * rd + 0] + acc[2][4 * rd + 0]; t1.x = acc[1][4 * rd + 0] - acc[2][4 * rd + 0] - acc[3][4 * rd + 0];
      t0.y = acc[0][4 * rd + 1] + acc[1][4 * rd + 1] + acc[2][4 * rd + 1]; t1.y = acc[1][4 * rd + 1] - acc[2][4 * rd + 1] - acc[3][4 * rd + 1];
      t0.z = acc[0][4 * rd + 2] + acc[1][4 * rd + 2] + acc[2][4 * rd + 2]; t1.z = acc[1][4 * rd + 2] - acc[2][4 * rd + 2] - acc[3][4 * rd + 2];
      t0.w = acc[0][4 * rd + 3] + acc[1][4 * rd + 3] + acc[2][4 * rd + 3]; t1.w = acc[1][4 * rd + 3] - acc[2][4 * rd + 3] - acc[3][4 * rd + 3];
      xw[(size_t)rd * (4 * 2 * 64)] = t0;
      xw[(size_t)rd * (4 * 2 * 64) + 64] = t1;
    }
  }
  __syncthreads();
  // Row stage: this wave finishes the channels 8 r + 4 h .. + 3 of its cout block (the accumulator group g4 = r of every
  // wave of the block): Y[0][b] = T0 + T1 + T2, Y[1][b] = T1 - T2 - T3.
  float4 y[2][2];
  {
    const float4* xr = reinterpret_cast<const float4*>(smem) + (((size_t)j * 4 + r) * 4) * 2 * 64 + lane;     // [j][dst = r][src][b][lane]
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      const float4 s0 = xr[(0 * 2 + bq) * 64], s1 = xr[(1 * 2 + bq) * 64], s2 = xr[(2 * 2 + bq) * 64], s3 = xr[(3 * 2 + bq) * 64];
      y[0][bq] = make_float4(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y, s0.z + s1.z + s2.z, s0.w + s1.w + s2.w);
      y[1][bq] = make_float4(s1.x - s2.x - s3.x, s1.y - s2.y - s3.y, s1.z - s2.z - s3.z, s1.w - s2.w - s3.w);
    }
  }

  // ---- bias, SiLU, residual, split, store: the lane's 2 x 2 pixels of tile mt, four channels each; lanes l and l + 32 hold
  // the two halves of the 8-channel group 4 j + r of the cout tile (conv_igemm_split.hip's epilogue, without its LDS pass) ----
  const float sc = P.acc_scale;
  const bool plain = P.out_plain != 0, act = P.act != 0;
  const void* const res_p = P.res;
  const int cg = ct * T::BN + 32 * j + 8 * r;         // first channel of the group
  const int mty = mt >> 3, mtx = mt & 7;
  bool sat = false;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
      const int oy = oy0 + 2 * mty + a, ox = ox0 + 2 * mtx + bq;
      const bool inside = oy < P.Ho && ox < P.Wo;
      const size_t pix = inside ? ((size_t)n * P.Ho + oy) * P.Wo + ox : 0;
      float2v v[2] = {__builtin_elementwise_fma(float2v{y[a][bq].x, y[a][bq].y}, float2v{sc, sc}, float2v{bias4.x, bias4.y}),
                      __builtin_elementwise_fma(float2v{y[a][bq].z, y[a][bq].w}, float2v{sc, sc}, float2v{bias4.z, bias4.w})};
      if (act) { v[0] = silu2(v[0]); v[1] = silu2(v[1]); }
      if (res_p) {                                   // uniform; the swaps need every lane
        uint4 rc = make_uint4(0, 0, 0, 0);           // lane l: the group's hi chunk, lane l + 32: its lo chunk
        if (inside) rc = *reinterpret_cast<const uint4*>(static_cast<const float*>(res_p) + pix * P.res_cstride + P.res_coff + cg + 4 * h);
        const auto sx = __builtin_amdgcn_permlane32_swap(rc.x, rc.z, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(rc.y, rc.w, false, false);
        const unsigned hw[2] = {sx[0], sy[0]}, lw[2] = {sx[1], sy[1]};
        const half4 rh = *reinterpret_cast<const half4*>(hw), rl = *reinterpret_cast<const half4*>(lw);
#pragma unroll
        for (int k = 0; k < 2; ++k)
          v[k] += float2v{(float)rh[2 * k], (float)rh[2 * k + 1]} + float2v{(float)rl[2 * k], (float)rl[2 * k + 1]};
      }
      float* dst = static_cast<float*>(P.out) + pix * P.out_cstride + P.out_coff + cg + 4 * h;
      if (plain) {
        if (inside) *reinterpret_cast<float4*>(dst) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
      } else {
        uint2 hi, lo;
        split2(v[0], hi.x, lo.x, sat);
        split2(v[1], hi.y, lo.y, sat);
        const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
        if (inside) *reinterpret_cast<uint4*>(dst) = make_uint4(sx[0], sy[0], sx[1], sy[1]);   // lane l: hi chunk at +0, lane l + 32: lo chunk at +16 bytes
      }
    }
  if (P.sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(P.sat_flag, 1);
}


// ------------------------------------------------------------------------------------------------------------------------
// Second form (ConvConfig::variant 4): 16 x 16 output pixels (8 x 8 Winograd tiles = TWO blocks of 32 MFMA columns) x 64 output
// channels per 256-thread workgroup, ONE wave per SIMD with the whole register file (512 registers per lane: 256 accumulators).
// What the first form's timing-only builds said it needs (profiles/r05_winograd_probe.txt): every weight fragment that comes in
// from L2 now feeds two column blocks (half the weight stream per output: the CU's 64 B/clk vector memory path was the floor),
// a thread transforms a WHOLE 4 x 4 tile of four channels (16 pixel reads and 320 vector instructions for 64 values: d = hi + lo is
// put together once per pixel instead of once per row combination), and every position's V fragments are read by one wave only.
// Wave w = transform row r: 4 positions x 2 column blocks x 2 cout blocks x 3 = 48 MFMAs per 16-channel chunk. Same packed
// weights as the first form. V is double-buffered (2 x 64 KB), the patch single-buffered (20.7 KB) behind a second barrier.
struct Wino2Tile {
  static constexpr int TH = 16, TW = 16, BN = 64, KC = 16;
  static constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;     // 18 x 18 input pixels
  static constexpr int RAW_UNITS = NPIX * 2;                         // 648 (pixel, 8-channel unit) pieces over 256 threads: 3 slots
  static constexpr int RAW_SLOTS = (RAW_UNITS + 255) / 256;
  static constexpr int RAW_BYTES = (NPIX + 1) * 64;                  // + the row nobody reads
  static constexpr int NT = 64;                                      // Winograd tiles: column blocks 0 (tile rows 0-3) and 1 (4-7)
  static constexpr int V_BYTES = 16 * NT * 64;                       // [position][tile][hi0 hi1 lo0 lo1]: 64 KB
  static constexpr int LOOP_BYTES = RAW_BYTES + 2 * V_BYTES;         // 151 872 B
  static constexpr int X_BYTES = 4 * 4 * 2 * 2 * 2 * 64 * 16;        // row-stage exchange [dst r][src r][tb][j][b][lane] float4: 128 KB
  static constexpr int LDS_BYTES = LOOP_BYTES > X_BYTES ? LOOP_BYTES : X_BYTES;
  static __host__ __device__ constexpr int raw_row(int py, int px) { return py * PW + (px & 1) * (PW / 2) + (px >> 1); }
  static __host__ __device__ constexpr int v_swz(int t) { return (2 * ((t >> 1) & 1)) ^ ((t >> 2) & 1); }
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_wino2_split_kernel(const ConvGroup g) {
  using T = Wino2Tile;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const lds_raw = smem;                        // [RAW_BYTES]
  char* const lds_v = smem + T::RAW_BYTES;           // [2][V_BYTES]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int r = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave = the transform row its MFMAs work on

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
  const int tx0 = pt % P.tiles_x;
  const int t2 = pt / P.tiles_x;
  const int ty0 = t2 % P.tiles_y + P.ty_first;
  const int n = t2 / P.tiles_y;
  const int oy0 = ty0 * T::TH, ox0 = tx0 * T::TW;
  const int nchunks = P.Cin / T::KC;
  const float* __restrict__ in = static_cast<const float*>(P.in);

  // ---- raw patch: up to three (pixel, 8-channel unit) pieces per thread; buffer loads, zero padding = an offset past the tensor ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((size_t)P.N * P.H * P.W * P.in_cstride * 4), 0x00020000);
  unsigned gvoff[T::RAW_SLOTS];
  int rdst[T::RAW_SLOTS];
#pragma unroll
  for (int s = 0; s < T::RAW_SLOTS; ++s) {
    const int u = tid + 256 * s;
    gvoff[s] = 0x80000000u;
    rdst[s] = T::NPIX * 64;
    if (u < T::RAW_UNITS) {
      const int pidx = u >> 1, c = u & 1;
      const int py = pidx / T::PW, px = pidx - py * T::PW;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      if (iy >= 0 && iy < P.H && ix >= 0 && ix < P.W) gvoff[s] = (unsigned)((((size_t)(n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff + c * 8) * 4);
      rdst[s] = T::raw_row(py, px) * 64 + c * 32;
    }
  }
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  uint4v ra_[T::RAW_SLOTS], rb_[T::RAW_SLOTS];
  const int last_chunk_off = (nchunks - 1) * (T::KC * 4);
#define GTXW2_RAW_LOAD(CHUNK)                                                                \
  {                                                                                          \
    const int so__ = min((CHUNK) * (T::KC * 4), last_chunk_off);                             \
    _Pragma("unroll") for (int s = 0; s < T::RAW_SLOTS; ++s) {                               \
      ra_[s] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gvoff[s], so__, 0);            \
      rb_[s] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gvoff[s], so__ + 16, 0);       \
    }                                                                                        \
  }
#define GTXW2_RAW_COMMIT()                                                                   \
  {                                                                                          \
    _Pragma("unroll") for (int s = 0; s < T::RAW_SLOTS; ++s) {                               \
      char* d__ = lds_raw + rdst[s];                                                         \
      *reinterpret_cast<uint4*>(d__) = make_uint4(ra_[s].x, ra_[s].y, rb_[s].x, rb_[s].y);   \
      *reinterpret_cast<uint4*>(d__ + 16) = make_uint4(ra_[s].z, ra_[s].w, rb_[s].z, rb_[s].w); \
    }                                                                                        \
  }

  // ---- weights: this wave's row r of a chunk = [j][c][hi | lo] = 16 fragments of 16 bytes per lane; reloaded position by position ----
  const uint4* __restrict__ usrc = reinterpret_cast<const uint4*>(P.wpack) + ((((size_t)ct * nchunks) * 4 + r) * 2) * (8 * 64) + lane;
  constexpr size_t U_CHUNK = 4 * 2 * 8 * 64;
  uint4 u[2][8];                                      // [j][2 c + hl]
#define GTXW2_U_LOAD(C, CHUNK)                                                               \
  {                                                                                          \
    const uint4* s__ = usrc + (size_t)(CHUNK) * U_CHUNK;                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                          \
      u[j][2 * (C)] = s__[(j * 8 + 2 * (C)) * 64];                                           \
      u[j][2 * (C) + 1] = s__[(j * 8 + 2 * (C) + 1) * 64];                                   \
    }                                                                                        \
  }

  // ---- transform job of this thread: tile tt (0..63), channel quad q: the whole 4 x 4 tile ----
  const int tt = tid >> 2, q = tid & 3;
  const int tty = tt >> 3, ttx = tt & 7;
  const float one = P.acc_scale * __builtin_amdgcn_rcpf(P.acc_scale);   // 1.0 the compiler cannot fold (keeps the fp16 -> fp32 fmas as v_fma_mix_f32)
  const int t_base = ((2 * tty) * T::PW + ttx) * 64 + q * 16;
  const int vsw_t = T::v_swz(tt & 31);
  const int vw_hi = tt * 64 + (((q >> 1) ^ vsw_t) << 4) + (q & 1) * 8;
  const int vw_lo = tt * 64 + (((2 + (q >> 1)) ^ vsw_t) << 4) + (q & 1) * 8;
  uint4 D__[4][4];                                    // the tile's 16 pixels (4 hi halves | 4 lo halves each), read behind the second barrier
  float2v Y__[4][2][4];                               // [input row][channel pair][position column]: the tile after its column stage
#define GTXW2_T_READ()                                                                       \
  {                                                                                          \
    _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                         \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                          \
        D__[rr][c] = *reinterpret_cast<const uint4*>(lds_raw + t_base + (rr * T::PW + (c & 1) * (T::PW / 2) + (c >> 1)) * 64); \
  }
  // Phase A, input rows RR0 and RR0 + 1: d = hi + lo for the row's four pixels and four channels, then the column stage
  // (d B: d0 - d2, d1 + d2, d2 - d1, d1 - d3). 32 v_fma_mix_f32 + 32 adds per call.
#define GTXW2_T_ROWS(RR0)                                                                    \
  {                                                                                          \
    _Pragma("unroll") for (int rr = (RR0); rr < (RR0) + 2; ++rr)                             \
      _Pragma("unroll") for (int cp = 0; cp < 2; ++cp) {                                     \
        float2v d__[4];                                                                      \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                      \
          const half2v h__ = __builtin_bit_cast(half2v, cp == 0 ? D__[rr][c].x : D__[rr][c].y); \
          const half2v l__ = __builtin_bit_cast(half2v, cp == 0 ? D__[rr][c].z : D__[rr][c].w); \
          d__[c] = float2v{__builtin_fmaf((float)l__[0], one, (float)h__[0]), __builtin_fmaf((float)l__[1], one, (float)h__[1])}; \
        }                                                                                    \
        Y__[rr][cp][0] = d__[0] - d__[2]; Y__[rr][cp][1] = d__[1] + d__[2];                  \
        Y__[rr][cp][2] = d__[2] - d__[1]; Y__[rr][cp][3] = d__[1] - d__[3];                  \
      }                                                                                      \
  }
  // Phase B, output rows R0 and R0 + 1 of the transformed tile: the row stage (Bt: y0 - y2, y1 + y2, y2 - y1, y1 - y3), the
  // split into hi + lo, and the 8-byte stores of the quad's halves to the row's four positions. 32 adds + 64 split instructions.
#define GTXW2_T_OUT(R0, VBUF)                                                                \
  {                                                                                          \
    _Pragma("unroll") for (int ro = (R0); ro < (R0) + 2; ++ro)                               \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                        \
        uint2 hi__, lo__;                                                                    \
        const float2v v0__ = ro == 0 ? Y__[0][0][c] - Y__[2][0][c] : (ro == 1 ? Y__[1][0][c] + Y__[2][0][c] : (ro == 2 ? Y__[2][0][c] - Y__[1][0][c] : Y__[1][0][c] - Y__[3][0][c])); \
        const float2v v1__ = ro == 0 ? Y__[0][1][c] - Y__[2][1][c] : (ro == 1 ? Y__[1][1][c] + Y__[2][1][c] : (ro == 2 ? Y__[2][1][c] - Y__[1][1][c] : Y__[1][1][c] - Y__[3][1][c])); \
        split2_mix(v0__, hi__.x, lo__.x, -one);                                              \
        split2_mix(v1__, hi__.y, lo__.y, -one);                                              \
        char* vp__ = lds_v + (VBUF) * T::V_BYTES + (4 * ro + c) * (T::NT * 64);              \
        *reinterpret_cast<uint2*>(vp__ + vw_hi) = hi__;                                      \
        *reinterpret_cast<uint2*>(vp__ + vw_lo) = lo__;                                      \
      }                                                                                      \
  }

  // ---- MFMA operand addresses: column block tb, column = lane & 31, k half h ----
  const int mt = lane & 31, h = lane >> 5;
  const int vsw_m = T::v_swz(mt);
  const int vr_hi = ((4 * r) * T::NT + mt) * 64 + ((h ^ vsw_m) << 4);
  const int vr_lo = ((4 * r) * T::NT + mt) * 64 + (((2 + h) ^ vsw_m) << 4);

  floatx16 acc[4][2][2];                              // [position c][column block][cout block]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][tb][j][i] = 0.f;

  // ---- prologue ----
  GTXW2_RAW_LOAD(0)
#pragma unroll
  for (int c = 0; c < 4; ++c) GTXW2_U_LOAD(c, 0)
  const float4 bias4[2] = {P.bias ? *reinterpret_cast<const float4*>(P.bias + ct * T::BN + 8 * r + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f),
                           P.bias ? *reinterpret_cast<const float4*>(P.bias + ct * T::BN + 32 + 8 * r + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f)};
  GTXW2_RAW_COMMIT()
  GTXW2_RAW_LOAD(1)
  __syncthreads();
  GTXW2_T_READ()
  GTXW2_T_ROWS(0) GTXW2_T_ROWS(2)
  GTXW2_T_OUT(0, 0) GTXW2_T_OUT(2, 0)
  __syncthreads();                                    // V(0) written, every read of the patch done
  GTXW2_RAW_COMMIT()
  GTXW2_RAW_LOAD(2)
  __syncthreads();
  if (nchunks > 1) GTXW2_T_READ()

  // One chunk as straight-line code: the MFMAs of chunk i (V buffer i & 1) position by position, behind each position's twelve the
  // reload of its weights for chunk i + 1 and a quarter of the transform of chunk i + 1 (its 16 pixels are in registers since the
  // end of the last chunk); then a barrier, the patch of chunk i + 2 from its registers into the (single) raw buffer, a barrier,
  // the next tile's 16 pixel reads. HAS_T: a chunk i + 1 exists; HAS_R: a chunk i + 2 exists.
// One transform unit per step: steps 0-7 = phase A for (input row, channel pair), steps 8-23 = phase B for (output row, column).
#define GTXW2_T_UNIT(S, VBUF)                                                                \
  {                                                                                          \
    if (GTXW_PROBE & 1) {                            /* timing only: the LDS traffic of the transform without its arithmetic */ \
      if ((S) >= 8) {                                                                        \
        const int ro = ((S) - 8) >> 2, c = ((S) - 8) & 3;                                    \
        char* vp__ = lds_v + (VBUF) * T::V_BYTES + (4 * ro + c) * (T::NT * 64);              \
        *reinterpret_cast<uint2*>(vp__ + vw_hi) = make_uint2(D__[ro][c].x, D__[ro][c].y);    \
        *reinterpret_cast<uint2*>(vp__ + vw_lo) = make_uint2(D__[ro][c].z, D__[ro][c].w);    \
      }                                                                                      \
    } else if ((S) < 8) {                                                                           \
      const int rr = (S) >> 1, cp = (S) & 1;                                                 \
      float2v d__[4];                                                                        \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                        \
        const half2v h__ = __builtin_bit_cast(half2v, cp == 0 ? D__[rr][c].x : D__[rr][c].y); \
        const half2v l__ = __builtin_bit_cast(half2v, cp == 0 ? D__[rr][c].z : D__[rr][c].w); \
        d__[c] = float2v{__builtin_fmaf((float)l__[0], one, (float)h__[0]), __builtin_fmaf((float)l__[1], one, (float)h__[1])}; \
      }                                                                                      \
      Y__[rr][cp][0] = d__[0] - d__[2]; Y__[rr][cp][1] = d__[1] + d__[2];                    \
      Y__[rr][cp][2] = d__[2] - d__[1]; Y__[rr][cp][3] = d__[1] - d__[3];                    \
    } else {                                                                                 \
      const int ro = ((S) - 8) >> 2, c = ((S) - 8) & 3;                                      \
      uint2 hi__, lo__;                                                                      \
      const float2v v0__ = ro == 0 ? Y__[0][0][c] - Y__[2][0][c] : (ro == 1 ? Y__[1][0][c] + Y__[2][0][c] : (ro == 2 ? Y__[2][0][c] - Y__[1][0][c] : Y__[1][0][c] - Y__[3][0][c])); \
      const float2v v1__ = ro == 0 ? Y__[0][1][c] - Y__[2][1][c] : (ro == 1 ? Y__[1][1][c] + Y__[2][1][c] : (ro == 2 ? Y__[2][1][c] - Y__[1][1][c] : Y__[1][1][c] - Y__[3][1][c])); \
      split2_mix(v0__, hi__.x, lo__.x, -one);                                                \
      split2_mix(v1__, hi__.y, lo__.y, -one);                                                \
      char* vp__ = lds_v + (VBUF) * T::V_BYTES + (4 * ro + c) * (T::NT * 64);                \
      *reinterpret_cast<uint2*>(vp__ + vw_hi) = hi__;                                        \
      *reinterpret_cast<uint2*>(vp__ + vw_lo) = lo__;                                        \
    }                                                                                        \
  }
  // MFMA number M of a chunk (0..47): position c = M / 12, cout block j, column block tb, term (lo x hi, hi x lo, hi x hi)
#define GTXW2_MFMA(M)                                                                        \
  {                                                                                          \
    const int c = (M) / 12, j = ((M) % 12) / 6, tb = (((M) % 12) % 6) / 3, term = (M) % 3;   \
    const half8 uh__ = *reinterpret_cast<const half8*>(&u[j][2 * c]);                        \
    const half8 ul__ = *reinterpret_cast<const half8*>(&u[j][2 * c + 1]);                    \
    if (!(GTXW_PROBE & 2)) acc[c][tb][j] = GTXW_MFMA(term == 0 ? ul__ : uh__, term == 1 ? bl__[c & 1][tb] : bh__[c & 1][tb], acc[c][tb][j]); \
    else if (term == 2) acc[c][tb][j][0] += (float)ul__[0] + (float)uh__[0] + (float)bl__[c & 1][tb][0] + (float)bh__[c & 1][tb][0]; \
  }
#define GTXW2_FRAGS(C)                                                                       \
  {                                                                                          \
    _Pragma("unroll") for (int tb = 0; tb < 2; ++tb) {                                       \
      bh__[(C) & 1][tb] = *reinterpret_cast<const half8*>(vh__ + (C) * (T::NT * 64) + tb * (32 * 64)); \
      bl__[(C) & 1][tb] = *reinterpret_cast<const half8*>(vl__ + (C) * (T::NT * 64) + tb * (32 * 64)); \
    }                                                                                        \
  }
  // One chunk in 24 pinned steps of two MFMAs and one transform unit each (one wave per SIMD: an MFMA holds the vector issue port
  // for 8 of its 32 cycles, so ~7 vector instructions of the transform of chunk i + 1 go into every gap instead of behind the
  // MFMAs; the scheduler only orders inside a step). Position c's V fragments are read three steps ahead of its first MFMA,
  // its weights for chunk i + 1 reloaded behind its last. Then a barrier, the patch of chunk i + 2 from its registers into the
  // (single) raw buffer, a barrier, the next tile's 16 pixel reads. HAS_T: a chunk i + 1 exists; HAS_R: a chunk i + 2 exists.
#define GTXW2_CHUNK(I, HAS_T, HAS_R)                                                         \
  {                                                                                          \
    const int b__ = (I) & 1;                                                                 \
    const char* vh__ = lds_v + b__ * T::V_BYTES + vr_hi;                                     \
    const char* vl__ = lds_v + b__ * T::V_BYTES + vr_lo;                                     \
    half8 bh__[2][2], bl__[2][2];                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    GTXW2_FRAGS(0)                                                                           \
    _Pragma("unroll") for (int st__ = 0; st__ < 24; ++st__) {                                   \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (st__ % 6 == 3 && st__ / 6 < 3) GTXW2_FRAGS(st__ / 6 + 1)                              \
      GTXW2_MFMA(2 * st__)                                                                    \
      GTXW2_MFMA(2 * st__ + 1)                                                                \
      if (HAS_T) {                                                                           \
        if (!(GTXW_PROBE & 4)) GTXW2_T_UNIT(st__, b__ ^ 1)                                    \
        if (st__ % 6 == 5 && !(GTXW_PROBE & 8)) GTXW2_U_LOAD(st__ / 6, (I) + 1)               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
      }                                                                                      \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    __syncthreads();                                  /* V(i + 1) is written; nobody reads the patch any more */ \
    if (HAS_R && !(GTXW_PROBE & 16)) {                                                       \
      GTXW2_RAW_COMMIT()                                                                     \
      GTXW2_RAW_LOAD((I) + 3)                                                                \
    }                                                                                        \
    if (HAS_R && !(GTXW_PROBE & 64)) __syncthreads();                                        \
    if (HAS_R) GTXW2_T_READ()                                                                \
  }
  int i = 0;
  for (; i + 2 < nchunks; ++i) GTXW2_CHUNK(i, true, true)
  if (i + 1 < nchunks) { GTXW2_CHUNK(i, true, false) ++i; }
  GTXW2_CHUNK(i, false, false)
#undef GTXW2_CHUNK
#undef GTXW2_FRAGS
#undef GTXW2_MFMA
#undef GTXW2_T_UNIT
#undef GTXW2_T_OUT
#undef GTXW2_T_ROWS
#undef GTXW2_T_READ
#undef GTXW2_U_LOAD
#undef GTXW2_RAW_COMMIT
#undef GTXW2_RAW_LOAD

  // ---- output transform: column stage in registers, row stage through LDS, every wave finishes the channel group g4 = its r ----
  {
    float4* xw = reinterpret_cast<float4*>(smem) + (size_t)r * (2 * 2 * 2 * 64) + lane;     // [dst][src = r][tb][j][b][lane]
#pragma unroll
    for (int rd = 0; rd < 4; ++rd)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float4 t0, t1;
#define GTXW2_COL(K, F) \
          t0.F = acc[0][tb][j][4 * rd + K] + acc[1][tb][j][4 * rd + K] + acc[2][tb][j][4 * rd + K]; \
          t1.F = acc[1][tb][j][4 * rd + K] - acc[2][tb][j][4 * rd + K] - acc[3][tb][j][4 * rd + K];
          GTXW2_COL(0, x) GTXW2_COL(1, y) GTXW2_COL(2, z) GTXW2_COL(3, w)
#undef GTXW2_COL
          float4* dst = xw + (size_t)rd * (4 * 2 * 2 * 2 * 64) + ((tb * 2 + j) * 2) * 64;
          dst[0] = t0;
          dst[64] = t1;
        }
  }
  __syncthreads();
  const float sc = P.acc_scale;
  const bool plain = P.out_plain != 0, act = P.act != 0;
  const void* const res_p = P.res;
  bool sat = false;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 y[2][2];
      const float4* xr = reinterpret_cast<const float4*>(smem) + (size_t)r * (4 * 2 * 2 * 2 * 64) + ((tb * 2 + j) * 2) * 64 + lane;   // [dst = r][src][tb][j][b][lane]
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        const float4 s0 = xr[0 * (2 * 2 * 2 * 64) + bq * 64], s1 = xr[1 * (2 * 2 * 2 * 64) + bq * 64], s2 = xr[2 * (2 * 2 * 2 * 64) + bq * 64],
                     s3 = xr[3 * (2 * 2 * 2 * 64) + bq * 64];
        y[0][bq] = make_float4(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y, s0.z + s1.z + s2.z, s0.w + s1.w + s2.w);
        y[1][bq] = make_float4(s1.x - s2.x - s3.x, s1.y - s2.y - s3.y, s1.z - s2.z - s3.z, s1.w - s2.w - s3.w);
      }
      const int cg = ct * T::BN + 32 * j + 8 * r;     // first channel of the lane pair's 8-channel group
      const int tile = tb * 32 + mt, mty = tile >> 3, mtx = tile & 7;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bq = 0; bq < 2; ++bq) {
          const int oy = oy0 + 2 * mty + a, ox = ox0 + 2 * mtx + bq;
          const bool inside = oy < P.Ho && ox < P.Wo;
          const size_t pix = inside ? ((size_t)n * P.Ho + oy) * P.Wo + ox : 0;
          float2v v[2] = {__builtin_elementwise_fma(float2v{y[a][bq].x, y[a][bq].y}, float2v{sc, sc}, float2v{bias4[j].x, bias4[j].y}),
                          __builtin_elementwise_fma(float2v{y[a][bq].z, y[a][bq].w}, float2v{sc, sc}, float2v{bias4[j].z, bias4[j].w})};
          if (act) { v[0] = silu2(v[0]); v[1] = silu2(v[1]); }
          if (res_p) {
            uint4 rc = make_uint4(0, 0, 0, 0);
            if (inside) rc = *reinterpret_cast<const uint4*>(static_cast<const float*>(res_p) + pix * P.res_cstride + P.res_coff + cg + 4 * h);
            const auto sx = __builtin_amdgcn_permlane32_swap(rc.x, rc.z, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(rc.y, rc.w, false, false);
            const unsigned hw[2] = {sx[0], sy[0]}, lw[2] = {sx[1], sy[1]};
            const half4 rh = *reinterpret_cast<const half4*>(hw), rl = *reinterpret_cast<const half4*>(lw);
#pragma unroll
            for (int k = 0; k < 2; ++k)
              v[k] += float2v{(float)rh[2 * k], (float)rh[2 * k + 1]} + float2v{(float)rl[2 * k], (float)rl[2 * k + 1]};
          }
          float* dst = static_cast<float*>(P.out) + pix * P.out_cstride + P.out_coff + cg + 4 * h;
          if (plain) {
            if (inside) *reinterpret_cast<float4*>(dst) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
          } else {
            uint2 hi, lo;
            split2(v[0], hi.x, lo.x, sat);
            split2(v[1], hi.y, lo.y, sat);
            const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
            if (inside) *reinterpret_cast<uint4*>(dst) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
    }
  if (P.sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(P.sat_flag, 1);
}

}  // namespace

// Packed image: [cout tile 64][cin chunk 16][row r][cout block j][position c][hi | lo][lane 64][8 halves] -- the A operand of
// v_mfma_f32_32x32x16_f16 as lane l holds it (row = cout 32 j + (l & 31), k = channels 8 (l >> 5) .. + 7 of the chunk) for
// U = G g Gt at position (r, c), G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]], computed in float64, scaled by
// the power of two that puts max |U| in [2^13, 2^14) and split into hi + lo; *acc_scale receives the inverse power of two.
std::vector<uint8_t> pack_conv_weights_wino(const float* w, int cout, int cin, const ConvConfig& cfg, float* acc_scale) {
  GTX_CHECK(cfg.ks == 3 && cfg.stride == 1 && cfg.bn == 64 && cfg.kc == 16 && cout % 64 == 0 && cin % 16 == 0,
            "conv (Winograd): 3x3 stride 1 with Cout a multiple of 64 and Cin a multiple of 16 only (Cin %d, Cout %d)", cin, cout);
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int n_ct = cout / 64, nchunks = cin / 16;
  std::vector<double> U((size_t)16 * cout * cin);
  double umax = 0.0;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      double gg[3][3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) gg[a][b] = w[((size_t)co * 9 + a * 3 + b) * cin + ci];
      for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
          double s = 0.0;
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) s += G[i][a] * gg[a][b] * G[k][b];
          U[((size_t)(i * 4 + k) * cout + co) * cin + ci] = s;
          umax = std::max(umax, std::fabs(s));
        }
    }
  int shift = 0;
  if (umax > 0.0 && std::isfinite(umax)) {
    int e;
    std::frexp(umax, &e);
    shift = std::max(-100, std::min(100, 14 - e));
  }
  *acc_scale = std::ldexp(1.f, -shift);
  std::vector<uint8_t> out((size_t)16 * cout * cin * 4);
  parallel_for(n_ct * nchunks, [&](int job) {
    const int ct = job / nchunks, ch = job % nchunks;
    for (int r = 0; r < 4; ++r)
      for (int j = 0; j < 2; ++j)
        for (int c = 0; c < 4; ++c)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = ct * 64 + 32 * j + (lane & 31);
            const size_t base = ((((((size_t)ct * nchunks + ch) * 4 + r) * 2 + j) * 4 + c) * 2) * 64 + lane;   // uint4 index of the hi fragment
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 16 + 8 * (lane >> 5) + e;
              const double v = std::ldexp(U[((size_t)(r * 4 + c) * cout + co) * cin + ci], shift);
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (double)hi);
              memcpy(&out[base * 16 + e * 2], &hi, 2);
              memcpy(&out[(base + 64) * 16 + e * 2], &lo, 2);
            }
          }
  });
  return out;
}

void conv_wino_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t stream) {
  for (int i = 0; i < g.count; ++i) {
    const ConvProblem& p = g.p[i];
    GTX_CHECK(c.ks == 3 && c.stride == 1 && c.bn == 64 && c.kc == 16 && p.Cout % 64 == 0 && p.Cin % 16 == 0 && p.Ho == p.H && p.Wo == p.W &&
                  !p.post_w && !p.front_img && p.c_split == 0,
              "conv (Winograd): 3x3 stride 1, pad 1, Cout %% 64 == 0, Cin %% 16 == 0 only (Cin %d, Cout %d)", p.Cin, p.Cout);
  }
  if (c.th == 16) {                                   // variant 4: 16 x 16 pixels, 4 waves, one per SIMD
    auto kern2 = conv_wino2_split_kernel;
    static std::once_flag once2;
    std::call_once(once2, [&] {
      GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, Wino2Tile::LDS_BYTES));
    });
    hipLaunchKernelGGL(kern2, dim3(g.grid_blocks), dim3(256), Wino2Tile::LDS_BYTES, stream, g);
    GTX_HIP(hipGetLastError());
    return;
  }
  auto kern = conv_wino_split_kernel;
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WinoTile::LDS_BYTES));
  });
  hipLaunchKernelGGL(kern, dim3(g.grid_blocks), dim3(512), WinoTile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace gtx
