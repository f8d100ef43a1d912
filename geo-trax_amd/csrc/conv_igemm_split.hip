// fp32-grade implicit-GEMM Conv(+bias+SiLU[+residual]) on the fp16 matrix pipe: "split-f16x3".
// gfx950 only. Activations are fp32-grade values (ultralytics.half: false, the reference default,
// geotrax/cfg/default.yaml:245) carried as two fp16 numbers,
//     x = hi + lo,   hi = fp16(x),   lo = fp16(x - hi)          (x - hi is exact in fp32)
// and they are stored in HBM ALREADY SPLIT ("pair format", split_format.hpp): the 32 bytes that would hold 8 fp32
// channels of a pixel hold their 8 hi halves followed by their 8 lo halves. The split is done once, by the epilogue that
// produces a value, instead of every time a consumer stages it (once per cout tile and per halo copy, 2 330 of 8 755
// cycles per K chunk in round 2's kernel): staging is now a straight 16-byte copy into the LDS row.
// A product w*x is formed by three fp16 MFMAs with fp32 accumulation,
//     w*x ~= w_hi*x_hi + w_hi*x_lo + w_lo*x_hi                   (the lo*lo term is < 2^-22 relative)
// i.e. 22 significand bits per operand instead of fp32's 24, at 3/16 of the cost of the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32 runs at 1/16 of the fp16 rate, MI355X_MICROARCH.md "Matrix cores").
// Weights are split on the host (pack_conv_weights_split) after an exact per-layer power-of-two scaling that
// puts max|w| near 2^13, so that w_lo stays in fp16's normal range; the epilogue multiplies the accumulator by
// the inverse power of two (exact). Activations are not scaled: x_lo is subnormal for |x| < 2^-3, where the
// absolute error floor 2^-25 is below fp32's own rounding error for |x| >= 0.5. Values beyond +-65504 are
// clamped when they are split (an fp32 network never gets there after BN + SiLU; the clamp only keeps an
// out-of-range value from turning into inf - inf) and the launch raises ConvProblem::sat_flag so that the host can say so.
//
// Two optional stages wrap the K loop of the 3x3 stride-2 instantiation (YOLOv8's first layers as ONE launch): the front stage
// (FrontTile below: the layer's input patch is computed from the RGB0 image, i.e. the stem, instead of loaded) and the post
// stage (ConvProblem::post_w: a 1x1 convolution on the output tile before it is stored).
//
// Work decomposition and LDS staging are those of conv_igemm.hip (8x16 output pixels x 32*WN couts per
// 4-wave workgroup, per K chunk the input patch and the KS*KS weight taps staged once, taps read shifted
// fragments); an LDS row holds the CPR hi chunks of its 8*CPR channels followed by the CPR lo chunks.
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
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define GTXS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// Diagnostic builds (`make stamp`) force-include csrc/diag/conv_split_diag.hpp, which defines these two hooks as clock
// stamps around the K loop; the shipped object has none.
#ifndef GTXS_DIAG_LOOP_BEGIN
#define GTXS_DIAG_ENTRY()
#define GTXS_DIAG_LOOP_BEGIN()
#define GTXS_DIAG_LOOP_END()
#define GTXS_DIAG_EXIT()
#define GTXS_DIAG_PHASE(K)
#endif

template <int KS, int STRIDE, int WN, int CPR, int WM>
struct SplitTile {
  static constexpr int TH = 8 * WM, TW = 16;         // WM sub-tiles of 2 rows x 16 pixels per wave
  static constexpr int BN = 32 * WN;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int NPIX = PH * PW;
  static constexpr int NCH = 2 * CPR;                // 16-B chunks per LDS row: hi 0..CPR-1, lo CPR..2CPR-1
  static constexpr int RB = NCH * 16;                // bytes per LDS row (64 or 128)
  static constexpr int KC = 8 * CPR;                 // input channels per K chunk
  static constexpr int PATCH_UNITS = NPIX * CPR;     // one unit = 8 channels of one pixel = 32 B of fp32 in HBM
  static constexpr int PATCH_SLOTS = (PATCH_UNITS + 255) / 256;
  static constexpr int W_CHUNKS = KS * KS * BN * NCH;
  static constexpr int W_SLOTS = (W_CHUNKS + 255) / 256;
  static constexpr int PATCH_BYTES = NPIX * RB;
  static constexpr int STAGE_BYTES = PATCH_BYTES + KS * KS * BN * RB;
  static constexpr int EPI_PITCH = BN * 4 + 16;      // fp32 epilogue transpose: bytes per staged pixel row
  static constexpr int EPI_BYTES = 4 * 32 * EPI_PITCH;   // one 32-pixel sub-tile per wave at a time
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int ROWS_PER_BANKROW = 256 / RB;  // 4 (RB=64) or 2 (RB=128)
  static __host__ __device__ constexpr int swz(int row) { return (row / ROWS_PER_BANKROW) & (NCH - 1); }
};

// x * sigmoid(x) with v_exp_f32 and v_rcp_f32 (1 ulp each); hipcc expands __fdividef to a full IEEE division (10 instructions)
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// The epilogues work on PAIRS of values: gfx950 has packed fp32 multiply / add / fma (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32: two values per lane and issue slot) and a packed fp32 -> fp16 conversion, so SiLU + the hi / lo split cost
// 8.5 vector instructions per value instead of 12.5 -- the epilogue is the VALU-bound part of a workgroup's life. Same
// operations in the same order as the scalar forms (silu_f; hi = fp16(x), lo = fp16(x - hi)): the results are theirs bit for bit.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2v relu2(const float2v v) { return float2v{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)}; }   // ConvProblem::act == 2 (RT-DETR's HGNetv2 blocks)
__device__ __forceinline__ float2v silu2(const float2v v) {
  const float2v t = v * -1.44269504088896341f;                      // exp(-v) = exp2(-v log2 e): what __expf compiles to
  const float2v d = float2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.f;
  return v * float2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// 2 fp32 values -> their two hi halves and two lo halves (one register each); sat becomes true when a value had to be clamped
__device__ __forceinline__ void split2(const float2v v, unsigned& hi, unsigned& lo, bool& sat) {
  const float2v x = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
  sat |= x.x != v.x || x.y != v.y;                                  // also true for a NaN (it is clamped to -65504 by v_med3)
  const half2v h = __builtin_convertvector(x, half2v);
  const half2v l = __builtin_convertvector(x - __builtin_convertvector(h, float2v), half2v);   // x - hi is exact in fp32
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// 4 fp32 values -> their 4 hi halves and 4 lo halves (8 bytes each)
__device__ __forceinline__ void split4(const float2v (&v)[2], uint2& hi, uint2& lo, bool& sat) {
  split2(v[0], hi.x, lo.x, sat);
  split2(v[1], hi.y, lo.y, sat);
}

// Front stage (ConvProblem::front_img, 3x3 stride-2 launches): the layer's input is YOLOv8's stem, SiLU(conv 3x3 stride 2 of
// the RGB0 byte image), and the workgroup computes the 17 x 33 patch of it that its tile reads instead of loading it -- the
// stem's output (236 MB per two 1920 x 1920 frames, written once and read back once) never exists in HBM. Per 16-channel K
// chunk: the 35 x 67 image pixels under the patch sit in LDS as {3 hi halves, 0, 3 lo halves, 0} of byte / 255 (16 B per
// pixel, in the region the chunk's weight taps take over afterwards), a wave walks 16-pixel tiles of the patch with
// v_mfma_f32_16x16x16_f16 (K = 4 taps x RGB0, three k-steps for the nine taps, three MFMAs per product like everything on
// this path: D[cout 16][pixel 16]), and bias + SiLU + the hi / lo split put the lane's four channels straight into the
// patch row the K loop reads. Positions of the patch outside the stem's output are the 3x3 layer's zero padding.
struct FrontTile {
  static constexpr int PH = 17, PW = 33, NPIX = PH * PW;       // the stride-2 kernel's patch, in stem-output pixels
  static constexpr int IH = 2 * PH + 1, IW = 2 * PW + 1;       // image pixels under it: 35 x 67
  static constexpr int IN_PIX = IH * IW;
  static constexpr int IN_SLOTS = (IN_PIX + 255) / 256;        // image words a thread keeps (both chunks stage from them)
  static constexpr int IN_BYTES = IN_PIX * 16;
  static constexpr int TILES = (NPIX + 15) / 16;               // 36 tiles of 16 patch pixels, 9 per wave
  static constexpr int LUT_BYTES = 256 * 4;
};

template <int KS, int STRIDE, int WN, int CPR, int WM, int FRONT>
constexpr int split_lds_bytes() {
  using Tile = SplitTile<KS, STRIDE, WN, CPR, WM>;
  if (!FRONT) return Tile::LDS_BYTES;
  const int wreg = Tile::STAGE_BYTES - Tile::PATCH_BYTES > FrontTile::IN_BYTES ? Tile::STAGE_BYTES - Tile::PATCH_BYTES : FrontTile::IN_BYTES;
  const int stage = Tile::PATCH_BYTES + wreg;
  return (stage > Tile::EPI_BYTES ? stage : Tile::EPI_BYTES) + FrontTile::LUT_BYTES;
}

template <int KS, int STRIDE, int WN, int CPR, int WM>
constexpr int split_min_waves() {
  if (WM == 2) return 2;                      // 64 accumulator + 64 fragment registers
  if (KS == 3) return STRIDE == 1 ? 3 : 2;    // stride 2: the 17x33 patch needs 5 prefetch slots of 8 registers
  return WN == 1 ? 5 : 4;
}

template <int KS, int STRIDE, int WN, int CPR, int WM, int FRONT>
__device__ __forceinline__ void conv_split_body(const ConvGroup& g) {
  using Tile = SplitTile<KS, STRIDE, WN, CPR, WM>;
  static_assert(!FRONT || (KS == 3 && STRIDE == 2 && WM == 1 && CPR == 2), "front stage: 3x3 stride 2, 16-channel chunks");
  constexpr int TH = Tile::TH, TW = Tile::TW, BN = Tile::BN, PW = Tile::PW, RB = Tile::RB, KC = Tile::KC;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + Tile::PATCH_BYTES;

  GTXS_DIAG_ENTRY()
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // launch header first, as one burst of scalar loads: group size and every member's first block
  const int cnt = g.count;
  int bb[kMaxGroup];
#pragma unroll
  for (int i = 0; i < kMaxGroup; ++i) bb[i] = g.p[i].block_begin;
  // XCD-aware logical block id: blocks b and b+8 share an XCD (speed only, never correctness); every XCD works through one
  // contiguous range of logical blocks -- the cout tiles of one pixel tile (same input patch) and neighbouring pixel tiles
  // (shared halo) meet in one L2 -- and the ranges hold equal work (ConvGroup::xcd_begin). Surplus blocks of the shorter
  // ranges leave here.
  const int xcd = blockIdx.x & 7;
  const int L = g.xcd_begin[xcd] + (int)(blockIdx.x >> 3);
  if (L >= g.xcd_begin[xcd + 1]) return;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < cnt && L >= bb[i]) pi = i;
  const ConvProblem P = g.p[pi];            // by value: one burst of wide scalar loads instead of a load (and a wait) per field

  const int lb = L - P.block_begin;
  const int ct = lb % P.n_ct;
  const int pt = lb / P.n_ct;
  const int tx = pt % P.tiles_x;
  const int t2 = pt / P.tiles_x;
  const int ty = t2 % P.tiles_y + P.ty_first;
  const int n = t2 / P.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * STRIDE - Tile::PAD, ix0 = ox0 * STRIDE - Tile::PAD;

  const float* __restrict__ in = static_cast<const float*>(P.in);
  const float* __restrict__ in2 = static_cast<const float*>(P.in2);
  const int nchunks = P.Cin / KC;
  constexpr int c_begin = 0;
  const int c_end = nchunks;

  long goff[Tile::PATCH_SLOTS];   // element offset of the unit (8 channels = 32 B: hi chunk, lo chunk), -1 = zero fill
  long goff2[Tile::PATCH_SLOTS];  // 1x1 only: the same unit in the half-resolution second source (ConvProblem::in2)
  int loff[Tile::PATCH_SLOTS];    // LDS byte offset of the unit's hi chunk, -1 = slot unused
  int lchk[Tile::PATCH_SLOTS];    // ... and of its lo chunk
  if constexpr (FRONT == 0) {
#pragma unroll
  for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {
    const int qid = tid + 256 * s;
    const int p = qid / CPR, c = qid % CPR;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool used = qid < Tile::PATCH_UNITS;
    const bool inb = used && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    goff[s] = inb ? ((long)(n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff + c * 8 : -1;
    goff2[s] = (KS == 1 && inb) ? ((long)(n * (P.H >> 1) + (iy >> 1)) * (P.W >> 1) + (ix >> 1)) * P.in2_cstride + P.in2_coff + c * 8 : -1;
    loff[s] = used ? p * RB + ((c ^ Tile::swz(p)) << 4) : -1;
    lchk[s] = used ? p * RB + (((CPR + c) ^ Tile::swz(p)) << 4) : -1;
  }
  }
  const uint4* __restrict__ wsrc =
      reinterpret_cast<const uint4*>(P.wpack) + (size_t)ct * nchunks * Tile::W_CHUNKS + tid;

  const int prow = lane & 31, h = lane >> 5;
  const int tcol = prow & 15;
  // sub-tile m of this wave: tile rows 2*(WM*wave + m) and +1
  const int trow0 = 2 * WM * wave + (prow >> 4);
  const int p0 = trow0 * STRIDE * PW + tcol * STRIDE;
  constexpr int PSUB = 2 * STRIDE * PW;              // patch rows between two sub-tiles

  uint4 pre_a[Tile::PATCH_SLOTS], pre_b[Tile::PATCH_SLOTS];
  uint4 pre_w[Tile::W_SLOTS];
#define GTXS_PREFETCH(CHUNK)                                                                 \
  {                                                                                          \
    const int c0__ = (CHUNK) * KC;                                                           \
    const bool up__ = KS == 1 && c0__ < P.c_split;   /* uniform: this chunk's channels come from the upsampled source */ \
    if constexpr (FRONT == 0) {                          /* front stage: the patch is computed, not loaded */ \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      uint4 va__ = make_uint4(0, 0, 0, 0), vb__ = make_uint4(0, 0, 0, 0);                    \
      if (goff[s] >= 0) {                                                                    \
        const uint4* src__ = reinterpret_cast<const uint4*>(up__ ? in2 + goff2[s] + c0__ : in + goff[s] + c0__); \
        va__ = src__[0];                                                                     \
        vb__ = src__[1];                                                                     \
      }                                                                                      \
      pre_a[s] = va__;                                                                       \
      pre_b[s] = vb__;                                                                       \
    }                                                                                        \
    }                                                                                        \
    const uint4* w__ = wsrc + (size_t)(CHUNK) * Tile::W_CHUNKS;                              \
    _Pragma("unroll") for (int s = 0; s < Tile::W_SLOTS; ++s) {                              \
      uint4 v__ = make_uint4(0, 0, 0, 0);                                                    \
      if (Tile::W_CHUNKS % 256 == 0 || tid + 256 * s < Tile::W_CHUNKS) v__ = w__[256 * s];  \
      pre_w[s] = v__;                                                                        \
    }                                                                                        \
  }
#define GTXS_COMMIT()                                                                        \
  {                                                                                          \
    if constexpr (FRONT == 0) {                                                                  \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      if (loff[s] >= 0) {                              /* pair format: the two halves of the unit ARE the LDS chunks */ \
        *reinterpret_cast<uint4*>(lds_patch + loff[s]) = pre_a[s];                           \
        *reinterpret_cast<uint4*>(lds_patch + lchk[s]) = pre_b[s];                           \
      }                                                                                      \
    }                                                                                        \
    }                                                                                        \
    _Pragma("unroll") for (int s = 0; s < Tile::W_SLOTS; ++s) {                              \
      if (Tile::W_CHUNKS % 256 == 0 || tid + 256 * s < Tile::W_CHUNKS)                       \
        *reinterpret_cast<uint4*>(lds_w + (tid + 256 * s) * 16) = pre_w[s];                  \
    }                                                                                        \
  }

  // ---- front stage: what a thread keeps for it (FRONT = the stem's 16-channel groups = this launch's K chunks, 1 or 2) ----
  constexpr int FN = FRONT > 0 ? FRONT : 1;
  unsigned f_raw[FrontTile::IN_SLOTS];              // its image pixels (RGB0 words)
  uint2 f_wh[FN][3], f_wl[FN][3];                   // stem weights: chunk, k-step s, this lane's (cout, tap) 4 halves
  float4 f_bias[FN];                                // ... and the bias of the lane's 4 output channels per chunk
  uint2 f_keep[FN > 1 ? FrontTile::TILES / 4 : 1][2];   // second chunk's patch values (hi, lo) of the lane, until the first chunk's matrix phase is over
  char* const f_stage = lds_w;                      // the image patch shares the weight taps' region
  unsigned* const f_lut = reinterpret_cast<unsigned*>(smem + split_lds_bytes<KS, STRIDE, WN, CPR, WM, FRONT>() - FrontTile::LUT_BYTES);
  int f_toff[3];                                    // byte offset of the lane's tap inside the staged image, per k-step
  if constexpr (FRONT > 0) {
    const unsigned* __restrict__ img = static_cast<const unsigned*>(P.front_img) + (size_t)n * P.front_h * P.front_w_px;
    const int gy0 = 2 * iy0 - 1, gx0 = 2 * ix0 - 1;   // image pixel under (row 0, col 0) of the staged patch
#pragma unroll
    for (int s = 0; s < FrontTile::IN_SLOTS; ++s) {
      const int i = tid + 256 * s;
      const int r = i / FrontTile::IW, c = i - r * FrontTile::IW;
      const int gy = gy0 + r, gx = gx0 + c;
      // byte 0 = value 0.0 = the stem's zero padding
      f_raw[s] = (i < FrontTile::IN_PIX && gy >= 0 && gy < P.front_h && gx >= 0 && gx < P.front_w_px) ? img[gy * P.front_w_px + gx] : 0u;
    }
#pragma unroll
    for (int c = 0; c < FN; ++c) {
      const uint2* fw = static_cast<const uint2*>(P.front_w) + (size_t)c * (3 * 2 * 64) + lane;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        f_wh[c][s] = fw[(2 * s + 0) * 64];
        f_wl[c][s] = fw[(2 * s + 1) * 64];
      }
      f_bias[c] = *reinterpret_cast<const float4*>(P.front_bias + c * KC + 4 * (lane >> 4));
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = min(4 * s + (lane >> 4), 8);    // taps 9..11 carry zero weights: any staged pixel will do
      f_toff[s] = ((tap / 3) * FrontTile::IW + tap % 3) * 16;
    }
    {                                                  // byte -> hi | lo << 16 of byte / 255 (ultralytics' `im / 255`, already split)
      const float f = (float)tid / 255.f;
      const _Float16 hi = (_Float16)f, lo = (_Float16)(f - (float)hi);
      f_lut[tid] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    }
  }
  // bias + SiLU + split of one tile's accumulator (4 channels of one patch pixel) -> the lane's 8 bytes of hi and of lo halves
#define GTXS_FRONT_EPI(ACC, BIAS, HV, LV)                                                    \
  {                                                                                          \
    /* SiLU of a sum of 27 products with inputs in [0, 1]: inside fp16's range for any sane weights; split2's clamp only keeps \
       an absurd checkpoint from producing inf - inf (no flag: stem_split_kernel has none either) */ \
    float2v v__[2] = {silu2(__builtin_elementwise_fma(float2v{(ACC)[0], (ACC)[1]}, float2v{P.front_scale, P.front_scale}, float2v{(BIAS).x, (BIAS).y})), \
                      silu2(__builtin_elementwise_fma(float2v{(ACC)[2], (ACC)[3]}, float2v{P.front_scale, P.front_scale}, float2v{(BIAS).z, (BIAS).w}))}; \
    if (!in__) v__[0] = v__[1] = float2v{0.f, 0.f};                                          \
    bool sat__ = false;                                                                      \
    split4(v__, HV, LV, sat__);                                                              \
  }
  // LDS patch row position of the lane's (hi, lo) 8-byte pieces for tile t__ (lanes past the patch's last pixel recomputed
  // that pixel: same bytes, same address)
#define GTXS_FRONT_PUT(T, HV, LV)                                                            \
  {                                                                                          \
    const int qq__ = min((wave + 4 * (T)) * 16 + (lane & 15), FrontTile::NPIX - 1);          \
    const int cg__ = lane >> 4;                              /* the lane's channels 4 cg .. 4 cg + 3 of the chunk */ \
    char* row__ = lds_patch + qq__ * RB + (cg__ & 1) * 8;                                    \
    *reinterpret_cast<uint2*>(row__ + (((cg__ >> 1) ^ Tile::swz(qq__)) << 4)) = (HV);        \
    *reinterpret_cast<uint2*>(row__ + (((CPR + (cg__ >> 1)) ^ Tile::swz(qq__)) << 4)) = (LV); \
  }
  // The front stage proper, once per workgroup: image patch -> LDS (GTXS_FRONT_STAGE), then the stem for every patch pixel
  // (GTXS_FRONT_TILES); the first chunk's 16 channels go to lds_patch, the second chunk's wait in f_keep (the B operands --
  // the staged pixels -- are read once for both).
#define GTXS_FRONT_STAGE()                                                                   \
  {                                                                                          \
    unsigned e__[FrontTile::IN_SLOTS][3];           /* every table read is issued before the first record is packed */ \
    _Pragma("unroll") for (int s = 0; s < FrontTile::IN_SLOTS; ++s) {                        \
      const unsigned px__ = f_raw[s];                                                        \
      e__[s][0] = f_lut[px__ & 255u]; e__[s][1] = f_lut[(px__ >> 8) & 255u]; e__[s][2] = f_lut[(px__ >> 16) & 255u]; \
    }                                                                                        \
    _Pragma("unroll") for (int s = 0; s < FrontTile::IN_SLOTS; ++s) {                        \
      const int i__ = tid + 256 * s;                                                         \
      if (FrontTile::IN_PIX % 256 == 0 || i__ < FrontTile::IN_PIX)                           \
        *reinterpret_cast<uint4*>(f_stage + i__ * 16) =                                      \
            make_uint4((e__[s][0] & 0xffffu) | (e__[s][1] << 16), e__[s][2] & 0xffffu, (e__[s][0] >> 16) | (e__[s][1] & 0xffff0000u), e__[s][2] >> 16); \
    }                                                                                        \
    __syncthreads();                                                                         \
  }
#define GTXS_FRONT_X(T)                              /* tile T's staged pixels: one 16-byte read per k-step */ \
  {                                                                                          \
    const int qc__ = min((wave + 4 * (T)) * 16 + (lane & 15), FrontTile::NPIX - 1);   /* patch pixel of this lane's MFMA column */ \
    const int py__ = qc__ / FrontTile::PW, px__ = qc__ - py__ * FrontTile::PW;               \
    const char* src__ = f_stage + (2 * py__ * FrontTile::IW + 2 * px__) * 16;                \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) f_x[(T) % 3][s] = *reinterpret_cast<const uint4*>(src__ + f_toff[s]); \
    const int sy__ = iy0 + py__, sx__ = ix0 + px__;                                          \
    f_in[(T) % 3] = sy__ >= 0 && sy__ < P.H && sx__ >= 0 && sx__ < P.W;   /* else: the 3x3 layer's zero padding */ \
  }
#define GTXS_FRONT_M(T)                              /* tile T's 9 MFMAs per chunk */          \
  {                                                                                          \
    _Pragma("unroll") for (int c = 0; c < FN; ++c) f_a[(T) % 2][c] = floatx4{0.f, 0.f, 0.f, 0.f}; \
    _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                          \
      const uint4 x__ = f_x[(T) % 3][s];                                                     \
      const uint2 xh2__ = make_uint2(x__.x, x__.y), xl2__ = make_uint2(x__.z, x__.w);        \
      const half4 xh__ = *reinterpret_cast<const half4*>(&xh2__), xl__ = *reinterpret_cast<const half4*>(&xl2__); \
      _Pragma("unroll") for (int c = 0; c < FN; ++c) {                                       \
        const half4 wh__ = *reinterpret_cast<const half4*>(&f_wh[c][s]), wl__ = *reinterpret_cast<const half4*>(&f_wl[c][s]); \
        f_a[(T) % 2][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl__, xh__, f_a[(T) % 2][c], 0, 0, 0); \
        f_a[(T) % 2][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh__, xl__, f_a[(T) % 2][c], 0, 0, 0); \
        f_a[(T) % 2][c] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh__, xh__, f_a[(T) % 2][c], 0, 0, 0); \
      }                                                                                      \
    }                                                                                        \
  }
  // Software pipeline over a wave's 9 tiles: the LDS reads run two tiles, the MFMAs one tile ahead of the bias + SiLU + split
  // of the current one, and the scheduler is told to put four vector instructions of that epilogue behind every MFMA.
#define GTXS_FRONT_TILES()                                                                   \
  {                                                                                          \
    constexpr int NT__ = FrontTile::TILES / 4;                                               \
    uint4 f_x[3][3];                                                                         \
    bool f_in[3];                                                                            \
    floatx4 f_a[2][FN];                                                                      \
    GTXS_FRONT_X(0)                                                                          \
    GTXS_FRONT_X(1)                                                                          \
    GTXS_FRONT_M(0)                                                                          \
    _Pragma("unroll") for (int t__ = 0; t__ < NT__; ++t__) {                                 \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (t__ + 2 < NT__) GTXS_FRONT_X(t__ + 2)                                              \
      if (t__ + 1 < NT__) GTXS_FRONT_M(t__ + 1)                                              \
      const bool in__ = f_in[t__ % 3];                                                       \
      uint2 hv0__, lv0__;                                                                    \
      GTXS_FRONT_EPI(f_a[t__ % 2][0], f_bias[0], hv0__, lv0__)                               \
      GTXS_FRONT_PUT(t__, hv0__, lv0__)                                                      \
      if constexpr (FN > 1) GTXS_FRONT_EPI(f_a[t__ % 2][FN - 1], f_bias[FN - 1], f_keep[t__][0], f_keep[t__][1]) \
      if (t__ + 1 < NT__) {                                                                  \
        _Pragma("unroll") for (int i__ = 0; i__ < 9 * FN; ++i__) {                           \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                 \
        }                                                                                    \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
  }

  // The accumulators start at bias / acc_scale (acc_scale is a power of two: exact), so the epilogue is one multiply and
  // has no loads of its own: the bias fetch overlaps the first global -> LDS round trip instead of opening the epilogue.
  float4 b4[WN][4];
#define GTXS_LOAD_BIAS()                                                                     \
  {                                                                                          \
    _Pragma("unroll") for (int j = 0; j < WN; ++j)                                           \
      _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) b4[j][g4] = make_float4(0.f, 0.f, 0.f, 0.f); \
    if (P.bias) {                             /* one uniform branch, eight independent loads */ \
      const float* __restrict__ bias_p = P.bias + ct * BN + 4 * (lane >> 5);                 \
      _Pragma("unroll") for (int j = 0; j < WN; ++j)                                         \
        _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) b4[j][g4] = *reinterpret_cast<const float4*>(bias_p + 32 * j + 8 * g4);   /* bias arrays are padded to whole cout tiles */ \
    }                                                                                        \
  }
  if constexpr (FRONT > 0) {
    // Front stage first: the accumulators and the post stage's weights do not exist yet (register room for f_keep); the first
    // chunk's weight taps and the bias are requested between the two halves and arrive while the stem is computed. The K
    // loop's opening barrier is also the one after which the weight taps may overwrite the staged image.
    __syncthreads();                                // the byte table is complete
    GTXS_FRONT_STAGE()
    GTXS_PREFETCH(c_begin)
    GTXS_LOAD_BIAS()
    GTXS_FRONT_TILES()
  } else {
    GTXS_PREFETCH(c_begin)
    GTXS_LOAD_BIAS()
  }
#undef GTXS_LOAD_BIAS

  floatx16 acc[WM][WN];
  {
    const float inv_sc = __builtin_amdgcn_rcpf(P.acc_scale);     // exact: acc_scale is a power of two
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          acc[m][j][4 * g4 + 0] = b4[j][g4].x * inv_sc; acc[m][j][4 * g4 + 1] = b4[j][g4].y * inv_sc;
          acc[m][j][4 * g4 + 2] = b4[j][g4].z * inv_sc; acc[m][j][4 * g4 + 3] = b4[j][g4].w * inv_sc;
        }
  }

  constexpr int NSTEP = KS * KS * (CPR / 2);
  half8 bh[2][WM], bl[2][WM], ah[2][WN], al[2][WN];
#define GTXS_LOAD_FRAGS(STEP, SLOT)                                                            \
    {                                                                                          \
      const int tap__ = (STEP) / (CPR / 2), ks__ = (STEP) % (CPR / 2);                         \
      const int c__ = 2 * ks__ + h;                                                            \
      _Pragma("unroll") for (int m = 0; m < WM; ++m) {                                         \
        const int p__ = p0 + m * PSUB + (tap__ / KS) * PW + (tap__ % KS);                      \
        const char* pr__ = lds_patch + p__ * RB;                                               \
        bh[SLOT][m] = *reinterpret_cast<const half8*>(pr__ + ((c__ ^ Tile::swz(p__)) << 4));   \
        bl[SLOT][m] = *reinterpret_cast<const half8*>(pr__ + (((CPR + c__) ^ Tile::swz(p__)) << 4)); \
      }                                                                                        \
      _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                         \
        const int nrow__ = 32 * j + prow;                                                      \
        const char* wr__ = lds_w + (tap__ * BN + nrow__) * RB;                                 \
        ah[SLOT][j] = *reinterpret_cast<const half8*>(wr__ + ((c__ ^ Tile::swz(nrow__)) << 4)); \
        al[SLOT][j] = *reinterpret_cast<const half8*>(wr__ + (((CPR + c__) ^ Tile::swz(nrow__)) << 4)); \
      }                                                                                        \
    }
  // The matrix phase of one K chunk: fragment reads run one (tap, k-step) ahead of the 3 * WM * WN MFMAs that consume them.
#define GTXS_MATRIX_PHASE()                                                                    \
    GTXS_LOAD_FRAGS(0, 0)                                                                      \
    _Pragma("unroll") for (int st = 0; st < NSTEP; ++st) {                                     \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      if (st + 1 < NSTEP) {                                                                    \
        if (st & 1) GTXS_LOAD_FRAGS(st + 1, 0) else GTXS_LOAD_FRAGS(st + 1, 1)                 \
      }                                                                                        \
      _Pragma("unroll") for (int m = 0; m < WM; ++m)                                           \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                       \
          /* small terms first, then the leading one */                                       \
          acc[m][j] = GTXS_MFMA(al[st & 1][j], bh[st & 1][m], acc[m][j]); \
          acc[m][j] = GTXS_MFMA(ah[st & 1][j], bl[st & 1][m], acc[m][j]); \
          acc[m][j] = GTXS_MFMA(ah[st & 1][j], bh[st & 1][m], acc[m][j]); \
        }                                                                                      \
      /* one fragment read of the next step behind every MFMA of this one */                   \
      _Pragma("unroll") for (int i__ = 0; i__ < 3 * WM * WN; ++i__) {                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                    \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                     \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
    }
  // post stage (below): the 1x1 layer's weights (16 KB per workgroup at 64 channels, L2-resident) are requested here, in front
  // of the K loop, and wait in registers: after the loop their round trip would be exposed once per workgroup
  constexpr int W2_CHUNKS = BN * BN * 4 / 16, W2_SLOTS = (KS == 3 && STRIDE == 2 && WM == 1) ? (W2_CHUNKS + 255) / 256 : 1;
  uint4 w2r[W2_SLOTS];
  if constexpr (KS == 3 && STRIDE == 2 && WM == 1) {
    if (P.post_w != nullptr) {
#pragma unroll
      for (int i = 0; i < W2_SLOTS; ++i)
        w2r[i] = tid + 256 * i < W2_CHUNKS ? static_cast<const uint4*>(P.post_w)[tid + 256 * i] : make_uint4(0, 0, 0, 0);
    }
  }
  GTXS_DIAG_LOOP_BEGIN()
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    __syncthreads();   // previous chunk's fragment reads are done
    if constexpr (FRONT > 1) {
      if (chunk != c_begin) {                       // the second chunk's 16 channels of the patch
        _Pragma("unroll") for (int t = 0; t < FrontTile::TILES / 4; ++t) GTXS_FRONT_PUT(t, f_keep[t][0], f_keep[t][1])
      }
    }
    GTXS_COMMIT()
    __syncthreads();
    if (chunk + 1 < c_end) GTXS_PREFETCH(chunk + 1)
    GTXS_MATRIX_PHASE()
    GTXS_DIAG_PHASE(7)
  }
  GTXS_DIAG_LOOP_END()
#undef GTXS_MATRIX_PHASE
#undef GTXS_LOAD_FRAGS
#undef GTXS_FRONT_STAGE
#undef GTXS_FRONT_TILES
#undef GTXS_FRONT_M
#undef GTXS_FRONT_X
#undef GTXS_FRONT_PUT
#undef GTXS_FRONT_EPI

  // ---- post stage (ConvProblem::post_w; 3x3 stride 2, one cout tile = all channels): a 1x1 convolution on the tile ----
  // y = SiLU(acc * 2^-shift) is split and staged as pair rows in this wave's LDS area (what the epilogue would have stored),
  // the 1x1 weights come into LDS as the packed image of the stand-alone 1x1 kernel (32-channel chunks, 128-byte rows), and
  // the wave multiplies its own 32 pixels: same k-step order, same three MFMAs per product as the stand-alone launch, so the
  // result is that launch's bit for bit. The accumulators are reused; the epilogue below then runs on the 1x1 layer.
  bool post = false;
  if constexpr (KS == 3 && STRIDE == 2 && WM == 1) {
    if (P.post_w != nullptr) {
      post = true;
      constexpr int PITCH = Tile::EPI_PITCH;
      constexpr int W2_OFF = 4 * 32 * PITCH;                                      // behind the waves' staging rows
      static_assert(W2_OFF + W2_CHUNKS * 16 <= Tile::LDS_BYTES, "post stage: the 1x1 weights do not fit behind the staging rows");
      float4 b2[WN][4];
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          b2[j][g4] = P.post_bias ? *reinterpret_cast<const float4*>(P.post_bias + 32 * j + 8 * g4 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
      __syncthreads();                            // every wave is done with the staging buffers of the K loop
      char* stg = smem + wave * (32 * PITCH);
      bool sat_y = false;
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int cl = 32 * j + 8 * g4 + 4 * h;
          float2v v[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            v[q] = float2v{acc[0][j][4 * g4 + 2 * q], acc[0][j][4 * g4 + 2 * q + 1]} * P.acc_scale;
            if (P.act == 1) v[q] = silu2(v[q]); else if (P.act == 2) v[q] = relu2(v[q]);
          }
          uint2 hi, lo;
          split4(v, hi, lo, sat_y);
          const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
          *reinterpret_cast<uint4*>(stg + prow * PITCH + cl * 4) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      if (P.sat_flag && __builtin_amdgcn_ballot_w64(sat_y) != 0 && lane == 0) atomicOr(P.sat_flag, 1);
#pragma unroll
      for (int i = 0; i < W2_SLOTS; ++i)
        if (tid + 256 * i < W2_CHUNKS) *reinterpret_cast<uint4*>(smem + W2_OFF + (tid + 256 * i) * 16) = w2r[i];
      __syncthreads();                            // the 1x1 weights are in LDS (the y rows are wave-private)
      const float inv2 = __builtin_amdgcn_rcpf(P.post_scale);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          acc[0][j][4 * g4 + 0] = b2[j][g4].x * inv2; acc[0][j][4 * g4 + 1] = b2[j][g4].y * inv2;
          acc[0][j][4 * g4 + 2] = b2[j][g4].z * inv2; acc[0][j][4 * g4 + 3] = b2[j][g4].w * inv2;
        }
      constexpr int K2 = BN / 32;                 // 32-channel chunks of the 1x1 layer's K = BN
#pragma unroll
      for (int k2 = 0; k2 < K2; ++k2)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int grp = 2 * (2 * k2 + ks) + h;  // the 8-channel group this half-wave feeds to the k-step
          const half8 yh = *reinterpret_cast<const half8*>(stg + prow * PITCH + grp * 32);
          const half8 yl = *reinterpret_cast<const half8*>(stg + prow * PITCH + grp * 32 + 16);
#pragma unroll
          for (int j = 0; j < WN; ++j) {
            const int nrow = 32 * j + prow, sw = (nrow >> 1) & 7, ci = 2 * ks + h;
            const char* wr = smem + W2_OFF + (k2 * BN + nrow) * 128;
            const half8 wh = *reinterpret_cast<const half8*>(wr + ((ci ^ sw) << 4));
            const half8 wl = *reinterpret_cast<const half8*>(wr + (((4 + ci) ^ sw) << 4));
            acc[0][j] = GTXS_MFMA(wl, yh, acc[0][j]);
            acc[0][j] = GTXS_MFMA(wh, yl, acc[0][j]);
            acc[0][j] = GTXS_MFMA(wh, yh, acc[0][j]);
          }
        }
    }
  }

  // ---- epilogue: acc * 2^-shift (bias is already in) -> SiLU (+ residual) -> split into hi / lo -> NHWC pair format ----
  // After the MFMAs a lane holds 4 consecutive channels of one pixel; lanes l and l + 32 hold the two halves of one
  // 8-channel group. Two v_permlane32_swap turn that into the group's 16-byte hi chunk (lane l) and 16-byte lo chunk
  // (lane l + 32), which land in the LDS staging row at the byte offset the fp32 float4 would have had; the wave then
  // stores whole BN*4-byte runs per pixel. ConvProblem::out_plain keeps plain fp32 (the Detect head's last stage, read by
  // the decode kernels).
  const float sc = post ? P.post_scale : P.acc_scale;
  const int cvalid = P.Cout - ct * BN;          // < BN in a last cout tile that is half empty (Cout = 16, 48, 80 ...)
  const bool plain = P.out_plain != 0;
  const int act = post ? P.post_act : P.act;      // 0 none, 1 SiLU, 2 ReLU
  const void* const res_p = P.res;
  const int o_cstride = P.out_cstride, o_coff = P.out_coff;
  float* const o_base = static_cast<float*>(P.out);
  bool sat = false;
  __syncthreads();                              // every wave is done with the staging buffers
#pragma unroll
  for (int m = 0; m < WM; ++m) {
    const int trow = trow0 + 2 * m;
    const int oy = oy0 + trow, ox = ox0 + tcol;
    constexpr int PITCH = Tile::EPI_PITCH;
    char* stg = smem + wave * (32 * PITCH);     // wave-private: its own LDS writes are ordered before its reads
    const bool inside = oy < P.Ho && ox < P.Wo;
    const size_t pix = inside ? ((size_t)n * P.Ho + oy) * P.Wo + ox : 0;
    const float* __restrict__ res =
        (res_p && inside) ? static_cast<const float*>(res_p) + pix * P.res_cstride + P.res_coff + ct * BN : nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = 32 * j + 8 * g4 + 4 * h;
        float2v v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v[q] = float2v{acc[m][j][4 * g4 + 2 * q], acc[m][j][4 * g4 + 2 * q + 1]} * sc;
          if (act == 1) v[q] = silu2(v[q]); else if (act == 2) v[q] = relu2(v[q]);
        }
        if (res_p) {                               // uniform; the swaps below need every lane
          uint4 rc = make_uint4(0, 0, 0, 0);       // lane l: the group's hi chunk, lane l + 32: its lo chunk
          if (res && cl < cvalid) rc = *reinterpret_cast<const uint4*>(res + cl);
          const auto sx = __builtin_amdgcn_permlane32_swap(rc.x, rc.z, false, false);   // -> (hi, lo) of this lane's channels 0, 1
          const auto sy = __builtin_amdgcn_permlane32_swap(rc.y, rc.w, false, false);   // ... and 2, 3
          const unsigned hw[2] = {sx[0], sy[0]}, lw[2] = {sx[1], sy[1]};
          const half4 rh = *reinterpret_cast<const half4*>(hw), rl = *reinterpret_cast<const half4*>(lw);
#pragma unroll
          for (int q = 0; q < 2; ++q)                                                   // hi + lo is exact in fp32
            v[q] += float2v{(float)rh[2 * q], (float)rh[2 * q + 1]} + float2v{(float)rl[2 * q], (float)rl[2 * q + 1]};
        }
        if (plain) {
          *reinterpret_cast<float4*>(stg + prow * PITCH + cl * 4) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
        } else {
          uint2 hi, lo;
          split4(v, hi, lo, sat);
          const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
          *reinterpret_cast<uint4*>(stg + prow * PITCH + cl * 4) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
      }
    }
    constexpr int LPP = BN / 4;                 // lanes per pixel (16 B each)
    constexpr int PPI = 64 / LPP;               // pixels per store instruction
#pragma unroll
    for (int it = 0; it < 32 / PPI; ++it) {
      const int p = it * PPI + lane / LPP, q = lane % LPP;
      const int py = oy0 + 2 * (WM * wave + m) + (p >> 4), px = ox0 + (p & 15);
      const uint4 val = *reinterpret_cast<const uint4*>(stg + p * PITCH + q * 16);
      if (py < P.Ho && px < P.Wo && (q >> 1) * 8 < cvalid) {     // cvalid is a multiple of 16: whole groups
        float* dst = o_base + (((size_t)n * P.Ho + py) * P.Wo + px) * o_cstride + o_coff + ct * BN + q * 4;
        *reinterpret_cast<uint4*>(dst) = val;
      }
    }
  }
  if (P.sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(P.sat_flag, 1);
  GTXS_DIAG_EXIT()
}

template <int KS, int STRIDE, int WN, int CPR, int WM>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 256), amdgpu_waves_per_eu((split_min_waves<KS, STRIDE, WN, CPR, WM>()))))
void conv_igemm_split_kernel(const ConvGroup g) {
  conv_split_body<KS, STRIDE, WN, CPR, WM, 0>(g);
}

// model.0 (the stem) + model.1 (3x3 stride 2) [+ model.2.cv1, the post stage] as one launch: see FrontTile
template <int WN, int NCH>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 256), amdgpu_waves_per_eu(2)))
void conv_front_split_kernel(const ConvGroup g) {
  conv_split_body<3, 2, WN, 2, 1, NCH>(g);
}

template <int KS, int STRIDE, int WN, int CPR, int WM>
void launch_t(const ConvGroup& g, hipStream_t stream) {
  using Tile = SplitTile<KS, STRIDE, WN, CPR, WM>;
  auto kern = conv_igemm_split_kernel<KS, STRIDE, WN, CPR, WM>;
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
  });
  hipLaunchKernelGGL(kern, dim3(g.grid_blocks), dim3(256), Tile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

template <int WN, int NCH>
void launch_front_t(const ConvGroup& g, hipStream_t stream) {
  constexpr int lds = split_lds_bytes<3, 2, WN, 2, 1, NCH>();
  static_assert(FrontTile::TILES % 4 == 0 && 2 * lds <= 160 * 1024, "front stage: 9 patch tiles per wave, two workgroups per CU");
  auto kern = conv_front_split_kernel<WN, NCH>;
  // GTX_FRONT_LDS_PAD=bytes: measurement hook (round 6): requests that much LDS on top of what the kernel uses, to price what its
  // 74 KB footprint costs beside the other detector stream's workgroups (profiles/r06_front_lds.txt)
  static const int pad = [] { const char* e = getenv("GTX_FRONT_LDS_PAD"); return e ? std::max(0, std::min(atoi(e), 160 * 1024 - lds)) : 0; }();
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds + pad));
  });
  hipLaunchKernelGGL(kern, dim3(g.grid_blocks), dim3(256), lds + pad, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace

// [cout tile][cin chunk][tap][n][swizzled 16-B chunk: hi chunks, then lo chunks], fp16; *acc_scale = 2^-shift
std::vector<uint8_t> pack_conv_weights_split(const float* w, int cout, int cin, const ConvConfig& cfg, float* acc_scale) {
  const int cpr = cfg.kc / 8, nch = 2 * cpr, rb = nch * 16, rpb = 256 / rb, taps = cfg.ks * cfg.ks;
  const int n_ct = (cout + cfg.bn - 1) / cfg.bn, nchunks = cin / cfg.kc;     // rows past Cout in the last tile are zero weights
  float wmax = 0.f;
  const size_t nw = (size_t)cout * taps * cin;
  for (size_t i = 0; i < nw; ++i) wmax = std::max(wmax, std::fabs(w[i]));
  int shift = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e;
    std::frexp(wmax, &e);          // wmax = m * 2^e, m in [0.5, 1)
    shift = 14 - e;                // max |w| * 2^shift in [2^13, 2^14)
    shift = std::max(-100, std::min(100, shift));
  }
  const float up = std::ldexp(1.f, shift);
  *acc_scale = std::ldexp(1.f, -shift);
  std::vector<uint8_t> out((size_t)n_ct * cfg.bn * taps * cin * 4);
  parallel_for(n_ct * nchunks, [&](int job) {                   // a (cout tile, cin chunk) pair owns a contiguous range of `out`
    const int ct = job / nchunks, ch = job % nchunks;
    for (int tap = 0; tap < taps; ++tap)
      for (int n = 0; n < cfg.bn; ++n) {
        const int sw = (n / rpb) & (nch - 1);
        const size_t row16 = ((((size_t)ct * nchunks + ch) * taps + tap) * cfg.bn + n) * nch;
        for (int c = 0; c < cpr; ++c)
          for (int e = 0; e < 8; ++e) {
            const int ci = ch * cfg.kc + c * 8 + e;
            const float v = ct * cfg.bn + n < cout ? w[((size_t)(ct * cfg.bn + n) * taps + tap) * cin + ci] * up : 0.f;   // exact (power of two)
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            memcpy(&out[(row16 + (size_t)(c ^ sw)) * 16 + e * 2], &hi, 2);
            memcpy(&out[(row16 + (size_t)((cpr + c) ^ sw)) * 16 + e * 2], &lo, 2);
          }
      }
  });
  return out;
}

// Stem weights for the front stage: [chunk of 16 couts][k-step 3][hi | lo][lane 64][4 halves]; lane (cout = lane % 16,
// kg = lane / 16) holds tap 4 s + kg (taps 9..11: zeros), channels R, G, B, 0 -- the A operand of v_mfma_f32_16x16x16_f16.
// Scaled by the power of two that puts max |w| in [2^13, 2^14) like every split weight; *acc_scale receives its inverse.
std::vector<uint16_t> pack_front_weights_split(const float* w27 /*[27][c0], (tap*3+ch) major*/, int c0, float* acc_scale) {
  const int chunks = (c0 + 15) / 16;
  float wmax = 0.f;
  for (int i = 0; i < 27 * c0; ++i) wmax = std::max(wmax, std::fabs(w27[i]));
  int shift = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e;
    std::frexp(wmax, &e);
    shift = std::max(-100, std::min(100, 14 - e));
  }
  const float up = std::ldexp(1.f, shift);
  *acc_scale = std::ldexp(1.f, -shift);
  std::vector<uint16_t> out((size_t)chunks * 3 * 2 * 64 * 4, 0);
  for (int ch = 0; ch < chunks; ++ch)
    for (int s = 0; s < 3; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int co = ch * 16 + (lane & 15), tap = 4 * s + (lane >> 4);
        for (int c = 0; c < 3; ++c) {
          const float v = (tap < 9 && co < c0) ? w27[(size_t)(tap * 3 + c) * c0 + co] * up : 0.f;
          const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
          memcpy(&out[((((size_t)ch * 3 + s) * 2 + 0) * 64 + lane) * 4 + c], &hi, 2);
          memcpy(&out[((((size_t)ch * 3 + s) * 2 + 1) * 64 + lane) * 4 + c], &lo, 2);
        }
      }
  return out;
}

void conv_split_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t s) {
  for (int i = 0; i < g.count; ++i)
    GTX_CHECK(g.p[i].post_w == nullptr || (c.ks == 3 && c.stride == 2 && c.th == 8 && g.p[i].Cout == c.bn && !g.p[i].res),
              "conv: the fused 1x1 post stage needs a 3x3 stride-2 launch whose cout tile holds every channel (Cout %d, tile %d)", g.p[i].Cout, c.bn);
  const int cpr = c.kc / 8, wn = c.bn / 32, wm = c.th / 8;
  if (g.p[0].front_img != nullptr) {
    const ConvProblem& p = g.p[0];
    GTX_CHECK(g.count == 1 && c.ks == 3 && c.stride == 2 && cpr == 2 && wm == 1 && (wn == 1 || wn == 2) && p.Cout <= c.bn &&
                  p.front_w && p.front_bias && p.front_h == 2 * p.H && p.front_w_px == 2 * p.W,
              "conv: the front stage needs a single 3x3 stride-2 launch with one cout tile on a stem output of half the image size");
    if (wn == 1 && p.Cin == 16) return launch_front_t<1, 1>(g, s);     // YOLOv8n: 16 -> 32 channels
    if (wn == 2 && p.Cin == 32) return launch_front_t<2, 2>(g, s);     // YOLOv8s: 32 -> 64
    fail(-3, "conv: no front-stage kernel for %d -> %d channels", p.Cin, p.Cout);
  }
#define GTX_CASE(KS, ST, WN, CPR, WM) \
  if (c.ks == KS && c.stride == ST && wn == WN && cpr == CPR && wm == WM) return launch_t<KS, ST, WN, CPR, WM>(g, s);
  GTX_CASE(3, 1, 1, 2, 1) GTX_CASE(3, 1, 2, 2, 1) GTX_CASE(3, 2, 1, 2, 1) GTX_CASE(3, 2, 2, 2, 1)
  GTX_CASE(1, 1, 1, 4, 1) GTX_CASE(1, 1, 2, 4, 1) GTX_CASE(1, 1, 1, 2, 1) GTX_CASE(1, 1, 2, 2, 1)

#undef GTX_CASE
  fail(-3, "conv (split-f16x3): no kernel for ks=%d stride=%d bn=%d kc=%d th=%d", c.ks, c.stride, c.bn, c.kc, c.th);
}

}  // namespace gtx

