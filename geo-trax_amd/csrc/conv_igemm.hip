// Implicit-GEMM Conv(+bias+SiLU[+residual]) on MFMA with LDS-staged im2col patches.
// gfx950 only. See conv_igemm.hpp for the data layout.
//
// Work decomposition
//   workgroup (256 threads = 4 waves) -> 8x16 output pixels x BN output channels
//   wave w                            -> output rows {2w, 2w+1} (32 pixels) x BN channels
//   MFMA                              -> D[cout 32][pixel 32] += W[cout][k] * X[k][pixel]
//                                        (weights are the A operand so that every lane ends up
//                                        holding 4 *consecutive* channels of one pixel: 8/16-B
//                                        NHWC stores instead of 2-B ones)
// K loop: for each chunk of KC input channels the (TH*s+KS-1)x(TW*s+KS-1) input patch and the
// KS*KS weight taps of that chunk are staged in LDS once; the KS*KS taps then read their A/B
// fragments from LDS at shifted addresses (im2col never exists in memory). Global loads of
// chunk i+1 are issued into registers before the MFMAs of chunk i (register prefetch).
// LDS rows are 64 or 128 B; 16-B chunks are XOR-swizzled so that ds_read_b128 fragment reads of
// 16 consecutive rows hit 16 distinct 16-B slots of the 256-B bank row.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <mutex>

#include "conv_igemm.hpp"

namespace gtx {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<_Float16> {
  using frag_t = half8;
  static __device__ __forceinline__ void run(floatx16& acc, const frag_t& a, const frag_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  using frag_t = float4;
  // 32x32x2 f32 MFMA: lane (i = l&31, kk = l>>5) supplies A[i][kk] / B[kk][i]. A 16-B chunk
  // per lane therefore feeds four MFMAs; both operands use the same k permutation.
  static __device__ __forceinline__ void run(floatx16& acc, const frag_t& a, const frag_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
};

template <typename T, int KS, int STRIDE, int WN, int CPR>
struct ConvTile {
  static constexpr int TH = 8, TW = 16;
  static constexpr int BN = 32 * WN;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int NPIX = PH * PW;
  static constexpr int RB = CPR * 16;                // bytes per LDS row
  static constexpr int EPC = 16 / (int)sizeof(T);    // elements per 16-B chunk
  static constexpr int KC = CPR * EPC;               // input channels per K chunk
  static constexpr int PATCH_CHUNKS = NPIX * CPR;
  static constexpr int PATCH_SLOTS = (PATCH_CHUNKS + 255) / 256;
  static constexpr int W_CHUNKS = KS * KS * BN * CPR;
  static constexpr int W_SLOTS = (W_CHUNKS + 255) / 256;
  static constexpr int PATCH_BYTES = NPIX * RB;
  static constexpr int STAGE_BYTES = PATCH_BYTES + KS * KS * BN * RB;
  static constexpr int EPI_BYTES = sizeof(T) == 2 ? 4 * 32 * (BN * 2 + 16) : 0;   // fp16 epilogue transpose
  static constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  static constexpr int ROWS_PER_BANKROW = 256 / RB;  // 4 (RB=64) or 2 (RB=128)
  static __host__ __device__ constexpr int swz(int row) {
    return (row / ROWS_PER_BANKROW) & (CPR - 1);
  }
};

// x * sigmoid(x) with v_exp_f32 and v_rcp_f32 (1 ulp each); hipcc expands __fdividef to a full IEEE division (10 instructions)
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <typename T> __device__ __forceinline__ void store4(T* dst, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<_Float16>(_Float16* dst, const float (&v)[4]) {
  half4 h;
  h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
  *reinterpret_cast<half4*>(dst) = h;
}
template <> __device__ __forceinline__ void store4<float>(float* dst, const float (&v)[4]) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}
template <typename T> __device__ __forceinline__ void load4(const T* src, float (&v)[4]);
template <> __device__ __forceinline__ void load4<_Float16>(const _Float16* src, float (&v)[4]) {
  half4 h = *reinterpret_cast<const half4*>(src);
  v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
}
template <> __device__ __forceinline__ void load4<float>(const float* src, float (&v)[4]) {
  float4 f = *reinterpret_cast<const float4*>(src);
  v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
}

// Waves per SIMD the register allocator is asked to fit (VGPR + AGPR <= 512 / waves, 8-register granules).
// Several instantiations sit within two registers of a boundary (3x3 stride 2, BN 64: 96 vs 98 VGPRs + 32 AGPRs
// is 4 or 3 waves) and flip with unrelated edits when left to the default heuristics; these are the values
// each one reaches without scratch.
template <typename T, int KS, int STRIDE, int WN, int CPR>
constexpr int conv_min_waves() {
  if (KS == 3 && CPR == 2) return WN == 1 ? (STRIDE == 1 ? 6 : 4) : 4;
  if (KS == 3 && CPR == 4) return STRIDE == 1 ? (WN == 1 ? 4 : 3) : (WN == 1 ? 3 : 2);
  if (KS == 1 && CPR == 4) return WN == 1 ? 8 : 5;
  if (KS == 1 && CPR == 8) return WN == 1 ? 7 : 5;
  return 1;
}

template <typename T, int KS, int STRIDE, int WN, int CPR>
__global__ __attribute__((amdgpu_flat_work_group_size(1, 256), amdgpu_waves_per_eu((conv_min_waves<T, KS, STRIDE, WN, CPR>()))))
void conv_igemm_kernel(const ConvGroup g) {
  using Tile = ConvTile<T, KS, STRIDE, WN, CPR>;
  using frag_t = typename Mma<T>::frag_t;
  constexpr int TH = Tile::TH, TW = Tile::TW, BN = Tile::BN, PW = Tile::PW, RB = Tile::RB;
  constexpr int EPC = Tile::EPC, KC = Tile::KC;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_patch = smem;
  char* lds_w = smem + Tile::PATCH_BYTES;

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

  const T* __restrict__ in = static_cast<const T*>(P.in);
  const int nchunks = P.Cin / KC;

  // ---- per-thread staging slots (loop invariant over K chunks) ----
  long goff[Tile::PATCH_SLOTS];   // element offset of this slot's 16-B chunk, -1 = zero fill
  int loff[Tile::PATCH_SLOTS];    // LDS byte offset, -1 = slot unused
#pragma unroll
  for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {
    const int qid = tid + 256 * s;
    const int p = qid / CPR, c = qid % CPR;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool used = qid < Tile::PATCH_CHUNKS;
    const bool inb = used && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
    goff[s] = inb ? ((long)(n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff + c * EPC : -1;
    loff[s] = used ? p * RB + ((c ^ Tile::swz(p)) << 4) : -1;
  }
  const uint4* __restrict__ wsrc =
      reinterpret_cast<const uint4*>(P.wpack) + (size_t)ct * nchunks * Tile::W_CHUNKS + tid;

  uint4 pre_p[Tile::PATCH_SLOTS];
  uint4 pre_w[Tile::W_SLOTS];
  // Register prefetch of one K chunk (global -> VGPR) and its commit (VGPR -> LDS). Macros, not
  // lambdas: captured arrays defeat SROA and end up in scratch.
#define GTX_PREFETCH(CHUNK)                                                                  \
  {                                                                                          \
    const int c0__ = (CHUNK) * KC;                                                           \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      uint4 v__ = make_uint4(0, 0, 0, 0);                                                    \
      if (goff[s] >= 0) v__ = *reinterpret_cast<const uint4*>(in + goff[s] + c0__);          \
      pre_p[s] = v__;                                                                        \
    }                                                                                        \
    const uint4* w__ = wsrc + (size_t)(CHUNK) * Tile::W_CHUNKS;                              \
    _Pragma("unroll") for (int s = 0; s < Tile::W_SLOTS; ++s) {                              \
      uint4 v__ = make_uint4(0, 0, 0, 0);                                                    \
      if (Tile::W_CHUNKS % 256 == 0 || tid + 256 * s < Tile::W_CHUNKS) v__ = w__[256 * s];  \
      pre_w[s] = v__;                                                                        \
    }                                                                                        \
  }
#define GTX_COMMIT()                                                                         \
  {                                                                                          \
    _Pragma("unroll") for (int s = 0; s < Tile::PATCH_SLOTS; ++s) {                          \
      if (loff[s] >= 0) *reinterpret_cast<uint4*>(lds_patch + loff[s]) = pre_p[s];           \
    }                                                                                        \
    _Pragma("unroll") for (int s = 0; s < Tile::W_SLOTS; ++s) {                              \
      if (Tile::W_CHUNKS % 256 == 0 || tid + 256 * s < Tile::W_CHUNKS)                       \
        *reinterpret_cast<uint4*>(lds_w + (tid + 256 * s) * 16) = pre_w[s];                  \
    }                                                                                        \
  }

  // ---- fragment addressing ----
  const int prow = lane & 31, h = lane >> 5;
  const int trow = 2 * wave + (prow >> 4), tcol = prow & 15;
  const int p0 = trow * STRIDE * PW + tcol * STRIDE;

  // The accumulators start at the bias, so the epilogue has no loads of its own: the bias fetch overlaps the first
  // global -> LDS round trip instead of opening the epilogue.
  floatx16 acc[WN];
  {
    float4 b4[WN][4];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) b4[j][g4] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias) {
      const float* __restrict__ bias_p = P.bias + ct * BN + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) b4[j][g4] = *reinterpret_cast<const float4*>(bias_p + 32 * j + 8 * g4);   // bias arrays are padded to whole cout tiles
    }
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        acc[j][4 * g4 + 0] = b4[j][g4].x; acc[j][4 * g4 + 1] = b4[j][g4].y;
        acc[j][4 * g4 + 2] = b4[j][g4].z; acc[j][4 * g4 + 3] = b4[j][g4].w;
      }
  }

  GTX_PREFETCH(0)
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();   // previous chunk's fragment reads are done
    GTX_COMMIT()
    __syncthreads();
    if (chunk + 1 < nchunks) GTX_PREFETCH(chunk + 1)
    if constexpr (KS == 3 && STRIDE == 1) {
      // Fragment reads run one (tap, k-step) ahead of the MFMAs that consume them: a wave's MFMAs then issue
      // back to back instead of each waiting for its own ds_read (the compiler's default schedule).
      constexpr int NSTEP = KS * KS * (CPR / 2);
      frag_t bq[2], aq[2][WN];
#define GTX_LOAD_FRAGS(STEP, SLOT)                                                             \
      {                                                                                          \
        const int tap__ = (STEP) / (CPR / 2), ks__ = (STEP) % (CPR / 2);                         \
        const int p__ = p0 + (tap__ / KS) * PW + (tap__ % KS);                                   \
        const int c__ = 2 * ks__ + h;                                                            \
        bq[SLOT] = *reinterpret_cast<const frag_t*>(lds_patch + p__ * RB + ((c__ ^ Tile::swz(p__)) << 4)); \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                         \
          const int nrow__ = 32 * j + prow;                                                      \
          aq[SLOT][j] = *reinterpret_cast<const frag_t*>(                                        \
              lds_w + (tap__ * BN + nrow__) * RB + ((c__ ^ Tile::swz(nrow__)) << 4));            \
        }                                                                                        \
      }
      GTX_LOAD_FRAGS(0, 0)
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NSTEP) {
          if (st & 1) GTX_LOAD_FRAGS(st + 1, 0) else GTX_LOAD_FRAGS(st + 1, 1)
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) Mma<T>::run(acc[j], aq[st & 1][j], bq[st & 1]);
        // the reads of step st+1 go out one at a time between the MFMAs of step st (as in conv_igemm_split.hip) instead of
        // as a burst in front of them
        constexpr int NM = WN * (sizeof(T) == 2 ? 1 : 4), NR = 1 + WN;
#pragma unroll
        for (int i = 0; i < (NM > NR ? NM : NR); ++i) {
          if (i < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (i < NM) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef GTX_LOAD_FRAGS
    } else {
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int p = p0 + ky * PW + kx;
          const int tap = ky * KS + kx;
          const char* prow_ptr = lds_patch + p * RB;
          const int pswz = Tile::swz(p);
#pragma unroll
          for (int ks = 0; ks < CPR / 2; ++ks) {
            const int c = 2 * ks + h;
            const frag_t bfrag = *reinterpret_cast<const frag_t*>(prow_ptr + ((c ^ pswz) << 4));
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int nrow = 32 * j + prow;
              const frag_t afrag = *reinterpret_cast<const frag_t*>(
                  lds_w + (tap * BN + nrow) * RB + ((c ^ Tile::swz(nrow)) << 4));
              Mma<T>::run(acc[j], afrag, bfrag);
            }
          }
        }
      }
    }
  }

  // ---- epilogue: bias + SiLU (+ residual) -> NHWC store ----
  // After the MFMAs a lane holds 4 consecutive channels of one pixel (8 B in fp16). Storing those
  // directly makes 8 partial writes per 128-B line; in the HBM-bound layers L2 evicts such lines before
  // they are complete and the memory side sees up to 3x the bytes (rocprofv3 WRITE_SIZE). fp16 path:
  // each wave transposes its 32 pixels x BN channels through LDS and stores 16 B per lane, whole
  // lines per pixel. The values are the same bits either way (rounded to fp16 before staging).
  const int oy = oy0 + trow, ox = ox0 + tcol;
  const int cvalid = P.Cout - ct * BN;          // < BN in a last cout tile that is half empty (Cout = 16, 48, 80 ...)
  const bool wide = sizeof(T) == 2 && (P.out_cstride % 8) == 0 && (P.out_coff % 8) == 0;
  if (wide) {
    constexpr int PITCH = BN * 2 + 16;          // bytes per staged pixel row; +16 keeps ds_write_b64 conflict free
    __syncthreads();                            // every wave is done with the staging buffers
    char* stg = smem + wave * (32 * PITCH);
    const bool inside = oy < P.Ho && ox < P.Wo;
    const size_t pix = inside ? ((size_t)n * P.Ho + oy) * P.Wo + ox : 0;
    const T* __restrict__ res =
        (P.res && inside) ? static_cast<const T*>(P.res) + pix * P.res_cstride + P.res_coff + ct * BN : nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = 32 * j + 8 * g4 + 4 * h;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[j][4 * g4 + i];
          if (P.act == 1) v[i] = silu(v[i]); else if (P.act == 2) v[i] = fmaxf(v[i], 0.f);
        }
        if (res && cl < cvalid) {
          float rv[4];
          load4<T>(res + cl, rv);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += rv[i];
        }
        store4<T>(reinterpret_cast<T*>(stg + prow * PITCH) + cl, v);
      }
    }
    // wave-private buffer: the wave's own LDS writes are ordered before its reads (lgkmcnt)
    constexpr int LPP = BN / 8;                 // lanes per pixel (16 B each)
    constexpr int PPI = 64 / LPP;               // pixels per store instruction
#pragma unroll
    for (int it = 0; it < 32 / PPI; ++it) {
      const int p = it * PPI + lane / LPP, q = lane % LPP;
      const int py = oy0 + 2 * wave + (p >> 4), px = ox0 + (p & 15);
      const uint4 val = *reinterpret_cast<const uint4*>(stg + p * PITCH + q * 16);
      if (py < P.Ho && px < P.Wo && q * 8 < cvalid) {
        T* dst = static_cast<T*>(P.out) + (((size_t)n * P.Ho + py) * P.Wo + px) * P.out_cstride + P.out_coff + ct * BN + q * 8;
        *reinterpret_cast<uint4*>(dst) = val;
      }
    }
  } else if (oy < P.Ho && ox < P.Wo) {
    const size_t pix = ((size_t)n * P.Ho + oy) * P.Wo + ox;
    T* __restrict__ out = static_cast<T*>(P.out) + pix * P.out_cstride + P.out_coff + ct * BN;
    const T* __restrict__ res =
        P.res ? static_cast<const T*>(P.res) + pix * P.res_cstride + P.res_coff + ct * BN : nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = 32 * j + 8 * g4 + 4 * h;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[j][4 * g4 + i];
          if (P.act == 1) v[i] = silu(v[i]); else if (P.act == 2) v[i] = fmaxf(v[i], 0.f);
        }
        if (cl >= cvalid) continue;               // channels past Cout (4 per lane: Cout is a multiple of 16)
        if (res) {
          float rv[4];
          load4<T>(res + cl, rv);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += rv[i];
        }
        store4<T>(out + cl, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------

namespace {
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
}  // namespace

ConvConfig conv_pick_config(int dtype, int ks, int stride, int cin, int cout, int force_kc, int force_bn, long out_pixels) {
  ConvConfig c{};
  c.dtype = dtype;
  c.ks = ks;
  c.stride = stride;
  GTX_CHECK((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1),
            "conv: unsupported kernel %dx%d stride %d", ks, ks, stride);
  if (dtype == DT_F32S) {
    // 3x3: 16-channel chunks (patch 11.5 KB + weight taps 36.9 KB per stage at 64 couts); 1x1: 32-channel chunks
    c.variant = 2;
    c.kc = ks == 1 ? (cin % 32 == 0 ? 32 : 16) : 16;
    if (force_kc > 0) c.kc = force_kc;
    c.bn = (cout % 64 == 0) ? 64 : 32;
    if (force_bn > 0) c.bn = force_bn;
    c.th = 8; c.tw = 16;
    GTX_CHECK(cin % c.kc == 0, "conv: Cin=%d is not a multiple of the K chunk %d", cin, c.kc);
    GTX_CHECK(cout % 16 == 0, "conv: Cout=%d is not a multiple of 16", cout);   // a last cout tile may be half empty (yolov8 n / m / x widths)
    // 3x3 stride 1 with whole 64-cout tiles: the Winograd F(2x2, 3x3) form (conv_wino_split.hip) is built, bit-exactness-tested
    // and OFF: alone it is 1.05-1.35 x faster than the direct kernel on the deep K loops (Cin >= 256) and slower on the others,
    // and next to the engine's second stream it loses even there (one 512-thread workgroup takes a CU's registers: 962 vs 970
    // frames/s; profiles/r05_winograd_probe.txt). GTX_WINO=1: every eligible layer, GTX_WINO=2: Cin >= 256 only, GTX_WINO=3: the second
    // form (16 x 16 pixels, one wave per SIMD: 1.04 x alone, its per-workgroup fixed cost is what is left).
    if (ks == 3 && stride == 1 && c.bn == 64 && c.kc == 16 && cout % 64 == 0) {
#ifdef GTX_WITH_WINO
      const int mode = env_int("GTX_WINO", 0);
#else
      const int mode = 0;                     // the default libgtx.so carries only what runs: `make WINO=1` builds conv_wino_split.hip in
#endif
      if (mode == 1 || (mode == 2 && cin >= 256)) c.variant = 3;
      if (mode == 3) { c.variant = 4; c.th = 16; }      // the 16 x 16-pixel form (one wave per SIMD)
      // 4: the second form only where it used less CU-time than the direct kernel alone: launches that fill the chip for several
      // rounds of its workgroups (>= 1024 of them: 240 x 240 maps at batch 2) with at least 8 chunks of K
      if (mode == 4 && cin >= 128 && out_pixels * (cout / 64) >= 1024L * 256) { c.variant = 4; c.th = 16; }
    }
    // 3x3 stride 1, 64-cout tiles, Cin a multiple of 32: the v_mfma_f32_16x16x32_f16 form (conv_k32_split.hip). GTX_K32=0: off
    if (ks == 3 && stride == 1 && c.bn == 64 && c.variant == 2 && c.kc == 16 && cin % 32 == 0 && env_int("GTX_K32", 1) != 0) {
      c.variant = 5;
      c.kc = 32;
    }
    // 1x1 with whole 64-cout tiles and 32-channel chunks: the pointwise v_mfma_f32_16x16x32_f16 forms (conv_k32p_split.hip) are built
    // by `make K32P=1`, bit-comparison-tested and OFF: 0.97 x / 0.81 x of the 32x32x16 kernel on RT-DETR's 1x1 layers
    // (profiles/r06_k32p_probe.txt). GTX_K32P=1: pixels staged in LDS, 2: pixels straight into the operand registers. Same weight
    // image as variant 2, so a caller that needs what only the 32x32x16 kernel has (a second upsampled source, a fused stage) sets
    // the variant back to 2.
#ifdef GTX_WITH_K32P
    if (ks == 1 && c.variant == 2 && c.kc == 32 && c.bn == 64 && cout % 64 == 0 && env_int("GTX_K32P", 0) != 0) c.variant = 6;
#endif
    return c;
  }
  const int epc = dtype == DT_F16 ? 8 : 4;
  int cpr = 4;
  if (ks == 1 && cin % (8 * epc) == 0) cpr = 8;
  // 3x3 fp16: 16-channel chunks for the shallow layers (half the LDS -> twice the resident workgroups,
  // which matters when a tile has only 2-8 chunks to pipeline), 32-channel chunks for Cin > 128
  if (ks == 3 && dtype == DT_F16) cpr = env_int("GTX_CONV_CPR3", cin <= 128 ? 2 : 4);
  if (dtype == DT_F16 && cin % (cpr * epc) != 0) cpr = 2;    // widths that are multiples of 16 only (yolov8 n / m / x: 48, 80, 144, 400 ...)
  if (force_kc > 0) cpr = force_kc / epc;    // members of a grouped launch must share one instantiation
  c.kc = cpr * epc;
  c.bn = (cout % 64 == 0) ? 64 : 32;
  if (force_bn > 0) c.bn = force_bn;
  c.th = 8; c.tw = 16;
  GTX_CHECK(cin % c.kc == 0, "conv: Cin=%d is not a multiple of the K chunk %d", cin, c.kc);
  GTX_CHECK(cout % 16 == 0, "conv: Cout=%d is not a multiple of 16", cout);
  return c;
}

namespace {
inline uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
}  // namespace

#ifndef GTX_WITH_K32P
// conv_k32p_split.hip is not part of this build (closed with numbers, profiles/r06_k32p_probe.txt); conv_pick_config never selects variant 6 here
void conv_k32p_launch(const ConvGroup&, const ConvConfig&, hipStream_t) { fail(-3, "the pointwise 16x16x32 kernels are not in this build (make -C geo-trax_amd K32P=1)"); }
#endif

#ifndef GTX_WITH_WINO
// conv_wino_split.hip is not part of this build (closed with numbers: 1.00-1.04 x of the direct kernel alone, -0.8 % end to end,
// profiles/r05_winograd_probe.txt); conv_pick_config never selects variants 3 / 4 here
void conv_wino_launch(const ConvGroup&, const ConvConfig&, hipStream_t) { fail(-3, "the Winograd kernels are not in this build (make -C geo-trax_amd WINO=1)"); }
std::vector<uint8_t> pack_conv_weights_wino(const float*, int, int, const ConvConfig&, float*) { fail(-3, "the Winograd kernels are not in this build (make -C geo-trax_amd WINO=1)"); }
#endif

std::vector<uint8_t> pack_conv_weights(const float* w, int cout, int cin, const ConvConfig& cfg, float* acc_scale) {
  if (acc_scale) *acc_scale = 1.f;
  if (cfg.variant >= 2 && cfg.variant <= 6) {
    float sc = 1.f;
    std::vector<uint8_t> r = (cfg.variant == 3 || cfg.variant == 4) ? pack_conv_weights_wino(w, cout, cin, cfg, &sc) : pack_conv_weights_split(w, cout, cin, cfg, &sc);
    if (acc_scale) *acc_scale = sc;
    return r;
  }
  const int es = (int)dtype_size(cfg.dtype);
  const int epc = 16 / es;
  const int cpr = cfg.kc / epc;
  const int rb = cpr * 16;
  const int rows_per_bankrow = 256 / rb;
  const int taps = cfg.ks * cfg.ks;
  const int n_ct = (cout + cfg.bn - 1) / cfg.bn, nchunks = cin / cfg.kc;   // rows past Cout in the last tile are zero weights
  std::vector<uint8_t> out((size_t)n_ct * cfg.bn * taps * cin * es);
  for (int ct = 0; ct < n_ct; ++ct)
    for (int ch = 0; ch < nchunks; ++ch)
      for (int tap = 0; tap < taps; ++tap)
        for (int n = 0; n < cfg.bn; ++n)
          for (int c = 0; c < cpr; ++c) {
            const int cs = c ^ ((n / rows_per_bankrow) & (cpr - 1));
            const size_t dst16 = ((((size_t)ct * nchunks + ch) * taps + tap) * cfg.bn + n) * cpr + cs;
            for (int e = 0; e < epc; ++e) {
              const int ci = ch * cfg.kc + c * epc + e;
              const float v = ct * cfg.bn + n < cout ? w[((size_t)(ct * cfg.bn + n) * taps + tap) * cin + ci] : 0.f;
              if (cfg.dtype == DT_F16) {
                const uint16_t hb = f32_to_f16_bits(v);
                memcpy(&out[dst16 * 16 + e * 2], &hb, 2);
              } else {
                memcpy(&out[dst16 * 16 + e * 4], &v, 4);
              }
            }
          }
  return out;
}

void conv_group_finalize(ConvGroup& g, const ConvConfig& cfg) {
  int total = 0;
  for (int i = 0; i < g.count; ++i) {
    ConvProblem& p = g.p[i];
    p.n_ct = (p.Cout + cfg.bn - 1) / cfg.bn;
    p.block_begin = total;
    p.tiles_x = cdiv(p.Wo, cfg.tw);
    p.tiles_y = cdiv(p.Ho, cfg.th);
    if (p.ty_count > 0) {
      GTX_CHECK(p.ty_first >= 0 && p.ty_first + p.ty_count <= p.tiles_y, "conv: tile rows [%d, %d) of %d", p.ty_first, p.ty_first + p.ty_count, p.tiles_y);
      p.tiles_y = p.ty_count;
    } else {
      p.ty_first = 0;
    }
    total += p.N * p.tiles_x * p.tiles_y * p.n_ct;
  }
  g.total_blocks = total;
  // XCD ranges of equal work. Work of a block = its K depth (members share taps, chunk and tile): with equal counts the
  // detection head's first stage gave XCD 0 only 8-chunk blocks of the 240x240 level and XCD 7 all the 32-chunk blocks of
  // the 60x60 level, and the launch lasted as long as XCD 7 needed (471 us instead of 290).
  double W = 0.0;
  for (int i = 0; i < g.count; ++i) W += (double)((i + 1 < g.count ? g.p[i + 1].block_begin : total) - g.p[i].block_begin) * g.p[i].Cin;
  g.xcd_begin[0] = 0;
  g.xcd_begin[8] = total;
  for (int k = 1; k < 8; ++k) {
    const double target = W * k / 8.0;
    double cum = 0.0;
    int at = total;
    for (int i = 0; i < g.count; ++i) {
      const int n = (i + 1 < g.count ? g.p[i + 1].block_begin : total) - g.p[i].block_begin;
      const double w = (double)n * g.p[i].Cin;
      if (cum + w >= target) {
        at = g.p[i].block_begin + std::min(n, (int)std::floor((target - cum) / g.p[i].Cin + 0.5));
        break;
      }
      cum += w;
    }
    g.xcd_begin[k] = std::max(at, g.xcd_begin[k - 1]);
  }
  int longest = 0;
  for (int k = 0; k < 8; ++k) longest = std::max(longest, g.xcd_begin[k + 1] - g.xcd_begin[k]);
  g.grid_blocks = 8 * longest;
}

double conv_flops(const ConvProblem& p, int ks) {
  return 2.0 * p.N * p.Ho * p.Wo * (double)p.Cout * p.Cin * ks * ks;
}

namespace {
template <typename T, int KS, int STRIDE, int WN, int CPR>
void launch_t(const ConvGroup& g, hipStream_t stream) {
  using Tile = ConvTile<T, KS, STRIDE, WN, CPR>;
  auto kern = conv_igemm_kernel<T, KS, STRIDE, WN, CPR>;
  static std::once_flag once;     // detectors on several host threads launch the same instantiation
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
  });
  hipLaunchKernelGGL(kern, dim3(g.grid_blocks), dim3(256), Tile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

template <typename T>
void launch_dt(const ConvGroup& g, const ConvConfig& c, hipStream_t s) {
  const int cpr = c.kc * (int)sizeof(T) / 16;
  const int wn = c.bn / 32;
#define GTX_CASE(KS, ST, WN, CPR) \
  if (c.ks == KS && c.stride == ST && wn == WN && cpr == CPR) return launch_t<T, KS, ST, WN, CPR>(g, s);
  GTX_CASE(3, 1, 1, 4) GTX_CASE(3, 1, 2, 4) GTX_CASE(3, 1, 1, 2) GTX_CASE(3, 1, 2, 2) GTX_CASE(3, 2, 2, 2) GTX_CASE(3, 2, 1, 2)
  GTX_CASE(3, 2, 1, 4) GTX_CASE(3, 2, 2, 4)
  GTX_CASE(1, 1, 1, 4) GTX_CASE(1, 1, 2, 4)
  GTX_CASE(1, 1, 1, 8) GTX_CASE(1, 1, 2, 8)
  GTX_CASE(1, 1, 1, 2) GTX_CASE(1, 1, 2, 2)
#undef GTX_CASE
  fail(-3, "conv: no kernel for ks=%d stride=%d bn=%d kc=%d", c.ks, c.stride, c.bn, c.kc);
}
}  // namespace

void conv_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream) {
  GTX_CHECK(g.count >= 1 && g.count <= kMaxGroup, "conv: bad group size %d", g.count);
  if (g.total_blocks == 0) return;
  for (int i = 0; i < g.count; ++i) {
    const ConvProblem& p = g.p[i];
    if (p.c_split == 0) continue;
    GTX_CHECK(cfg.variant == 2 && cfg.ks == 1, "conv: a second (upsampled) source needs the split-f16x3 1x1 kernel");
    GTX_CHECK(p.in2 && p.c_split % cfg.kc == 0 && p.c_split < p.Cin && p.H % 2 == 0 && p.W % 2 == 0 && p.in2_cstride % 4 == 0 && p.in2_coff % 4 == 0,
              "conv: bad second source (c_split=%d, %dx%d)", p.c_split, p.W, p.H);
  }
  if (cfg.variant == 3 || cfg.variant == 4) return conv_wino_launch(g, cfg, stream);
  if (cfg.variant == 5) return conv_k32_launch(g, cfg, stream);
  if (cfg.variant == 6) return conv_k32p_launch(g, cfg, stream);
  if (cfg.variant == 2) return conv_split_launch(g, cfg, stream);
  if (cfg.dtype == DT_F16) launch_dt<_Float16>(g, cfg, stream);
  else launch_dt<float>(g, cfg, stream);
}

const char* conv_kernel_name(const ConvConfig& c) {
  static thread_local char buf[96];
  if (c.variant == 3) return "conv_wino_split_kernel";
  if (c.variant == 4) return "conv_wino2_split_kernel";
  if (c.variant == 5) return "conv_k32_split_kernel";
  if (c.variant == 6) return "conv_k32p_split_kernel";
  if (c.variant == 2) {
    snprintf(buf, sizeof buf, "conv_igemm_split_kernel<%d, %d, %d, %d, %d>", c.ks, c.stride, c.bn / 32, c.kc / 8, c.th / 8);
    return buf;
  }
  snprintf(buf, sizeof buf, "conv_igemm_kernel<%s, %d, %d, %d, %d>", c.dtype == DT_F16 ? "_Float16" : "float",
           c.ks, c.stride, c.bn / 32, c.kc * (int)dtype_size(c.dtype) / 16);
  return buf;
}

}  // namespace gtx
