// Implicit-GEMM Conv(+bias+SiLU[+residual]) on MFMA, second generation (fp16): LDS-DMA staging
// into a ring of stages, pixel blocks shaped for conflict-free fragment reads. gfx950 only.
// Same problem description and epilogue contract as conv_igemm.hip (see conv_igemm.hpp).
//
// Work decomposition
//   workgroup (4 waves) -> TH x (32*TWB) output pixels x BN = 32*NR output channels
//   pixel block         -> 32 consecutive pixels of one output row (1x1 convs: of the linearised
//                          N*H*W pixel index). Lane l of an MFMA B fragment then reads LDS row
//                          base + l: with the XOR swizzle below every ds_read_b128 is conflict
//                          free for any base (scratch/lds_conflict2.py enumerates the lane groups).
//   wave w              -> pixel blocks {w*MR .. w*MR+MR-1} x all NR channel blocks:
//                          MR*NR MFMA 32x32x16 per k-step from MR + NR fragment reads.
// K loop: one stage = KC = 8*CPR input channels of the (PH x PW) input patch plus the KS*KS weight
// taps of those channels. Stages are filled by global_load_lds_dwordx4 (no VGPR round trip; the
// LDS image is lane-linear, so the swizzle is applied to the per-lane SOURCE address and again on
// the fragment read) into a ring of NS slots; NS-1 stages are in flight while one is consumed:
//     s_waitcnt vmcnt(loads of the stages issued after this one) ; s_barrier ;
//     issue stage c+NS-1 into the slot consumed in iteration c-1 ; MFMAs of stage c.
// Out-of-image patch pixels read a zero page. Stride-2 patches are stored with even and odd
// columns de-interleaved so that a tap's 32 lanes still read consecutive LDS rows.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <type_traits>

#include "conv_igemm.hpp"

namespace gtx {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int round_up_c(int a, int b) { return (a + b - 1) / b * b; }

// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N)
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

template <int KS, int STRIDE, int TH, int TWB, int NR, int CPR, int NS>
struct Tile2 {
  static constexpr int PB = TH * TWB;              // pixel blocks per workgroup
  static constexpr int MR = PB / 4;                // pixel blocks per wave
  static constexpr int BN = 32 * NR;
  static constexpr int TW = 32 * TWB;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int PWE = (PW + 1) / 2;         // even columns (stride-2 de-interleave)
  static constexpr int NPIX = PH * PW;
  static constexpr int RB = CPR * 16;              // bytes per LDS row (one pixel / one cout, KC channels)
  static constexpr int KC = CPR * 8;
  static constexpr int RPB = 256 / RB;             // rows per 256-B bank row
  static constexpr int PATCH_BYTES = round_up_c(NPIX * RB, 256);
  static constexpr int W_ROWS = KS * KS * BN;
  static constexpr int W_BYTES = W_ROWS * RB;
  static constexpr int W_SLOTS = W_BYTES / 16;
  static constexpr int USED_SLOTS = (PATCH_BYTES + W_BYTES) / 16;
  static constexpr int NLD = (USED_SLOTS + 255) / 256;   // glds instructions per thread per stage
  static constexpr int STAGE_BYTES = NLD * 4096;
  static constexpr int LDS_BYTES = NS * STAGE_BYTES;
  static_assert(PB % 4 == 0, "pixel blocks must split over 4 waves");
  static_assert(CPR == 2 || CPR == 4 || CPR == 8, "row = 32, 64 or 128 bytes");
  static __host__ __device__ constexpr int swz(int row) { return (row / RPB) & (CPR - 1); }
};

__device__ __forceinline__ float silu2(float v) { return __fdividef(v, 1.f + __expf(-v)); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt lgkmcnt(N) that the fragments of the current step are tied to: their consumers cannot
// be scheduled above it.
template <int N, int MR, int NR>
__device__ __forceinline__ void wait_frags(half8 (&a)[NR], half8 (&b)[MR]) {
  static_assert(MR <= 2 && NR <= 4, "extend the operand lists");
  if constexpr (MR == 1 && NR == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(N));
  else if constexpr (MR == 1 && NR == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(N));
  else if constexpr (MR == 2 && NR == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  else if constexpr (MR == 2 && NR == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
  else if constexpr (MR == 1 && NR == 4) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]) : "n"(N));
  else if constexpr (MR == 2 && NR == 4) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
}

#define GTX_GLDS16(SRC, DST)                                                                    \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                   (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

template <int KS, int STRIDE, int TH, int TWB, int NR, int CPR, int NS>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(const ConvGroup g) {
  using T = _Float16;
  using Tile = Tile2<KS, STRIDE, TH, TWB, NR, CPR, NS>;
  constexpr int MR = Tile::MR, BN = Tile::BN, PW = Tile::PW, RB = Tile::RB, KC = Tile::KC, NLD = Tile::NLD;

  extern __shared__ __attribute__((aligned(256))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // XCD-aware logical block id (blocks b and b+8 share an XCD): contiguous runs per XCD so that the
  // cout tiles of one pixel tile and neighbouring pixel tiles meet in one L2.
  int L;
  {
    const int b = blockIdx.x, nb = g.total_blocks;
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kMaxGroup; ++i)
    if (i < g.count && L >= g.p[i].block_begin) pi = i;
  const ConvProblem& P = g.p[pi];

  const int lb = L - P.block_begin;
  const int ct = lb % P.n_ct;
  const int pt = lb / P.n_ct;
  const int tx = pt % P.tiles_x;
  const int t2 = pt / P.tiles_x;
  const int ty = t2 % P.tiles_y;
  const int n = t2 / P.tiles_y;
  // 1x1 convolutions run on the linearised pixel index: "image" = 1 row of N*H*W pixels
  const int imgH = KS == 1 ? 1 : P.H, imgW = KS == 1 ? P.N * P.H * P.W : P.W;
  const int outH = KS == 1 ? 1 : P.Ho, outW = KS == 1 ? imgW : P.Wo;
  const int oy0 = ty * TH, ox0 = tx * Tile::TW;
  const int iy0 = oy0 * STRIDE - Tile::PAD, ix0 = ox0 * STRIDE - Tile::PAD;

  const char* __restrict__ in = static_cast<const char*>(P.in);
  const int nchunks = P.Cin / KC;

  // ---- per-thread LDS-DMA slots: source pointer of chunk 0 and its per-chunk increment ----
  const char* src[NLD];
  int inc[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int q = i * 256 + tid;               // 16-B slot of the stage image this lane fills
    const char* p = static_cast<const char*>(P.zero);
    int step = 0;
    if (q * 16 < Tile::NPIX * RB) {
      const int r = q / CPR, cs = q % CPR;
      const int c = cs ^ Tile::swz(r);
      const int py = r / PW, rem = r - py * PW;
      int px = rem;
      if (STRIDE == 2) px = rem >= Tile::PWE ? 2 * (rem - Tile::PWE) + 1 : 2 * rem;
      const int iy = iy0 + py, ix = ix0 + px;
      if (iy >= 0 && iy < imgH && ix >= 0 && ix < imgW) {
        if (P.in_blocked) {
          p = in + ((((long)n * (P.in_cstride / 16) * imgH + iy) * imgW + ix) * 16 + (c & 1) * 8) * (long)sizeof(T) + (long)(c >> 1) * imgH * imgW * 32;
          step = (KC / 16) * imgH * imgW * 32;
        } else {
        p = in + ((((long)n * imgH + iy) * imgW + ix) * P.in_cstride + P.in_coff + c * 8) * (long)sizeof(T);
        step = KC * (int)sizeof(T);
        }
      }
    } else if (q * 16 >= Tile::PATCH_BYTES && q < Tile::USED_SLOTS) {
      p = static_cast<const char*>(P.wpack) + ((long)ct * nchunks * Tile::W_SLOTS + (q - Tile::PATCH_BYTES / 16)) * 16;
      step = Tile::W_BYTES;
    }
    src[i] = p;
    inc[i] = step;
  }
#define GTX_ISSUE_STAGE(SLOT)                                                     \
  {                                                                               \
    char* dst__ = smem + (SLOT) * Tile::STAGE_BYTES + wave * 1024;                \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                             \
      GTX_GLDS16(src[i], dst__ + i * 4096);                                       \
      src[i] += inc[i];                                                           \
    }                                                                             \
  }

  // ---- fragment addressing ----
  const int prow = lane & 31, h = lane >> 5;
  // B: LDS row of this lane for pixel block m, tap (ky,kx) = brow(m,ky,kx) + prow
  // A: row (tap*BN + 32 j + prow); the multiples of 32 do not change the swizzle term
  int a_off[CPR / 2];
#pragma unroll
  for (int ks = 0; ks < CPR / 2; ++ks) a_off[ks] = Tile::PATCH_BYTES + prow * RB + (((2 * ks + h) ^ Tile::swz(prow)) << 4);

  floatx16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;

  // prologue: NS-1 stages in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nchunks) GTX_ISSUE_STAGE(s)

  int slot = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    // stages issued after `chunk` that may still be in flight: min(NS-2, nchunks-1-chunk)
    {
      const int ahead = min(NS - 2, nchunks - 1 - chunk);
      if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * NLD>();
      else if (NS >= 3 && ahead >= 1) wait_vmcnt<NLD>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (chunk + NS - 1 < nchunks) {
      const int ns = slot == 0 ? NS - 1 : slot - 1;   // the slot consumed in the previous iteration
      GTX_ISSUE_STAGE(ns)
    }
    // Fragment reads run one step (= one tap x 16 channels) ahead of the MFMAs that consume them.
    // The reads and their counted waits are inline asm: hipcc's own s_waitcnt placement drains
    // lgkmcnt to 0 in front of every second MFMA group, which exposes the LDS latency again.
    constexpr int KSTEPS = CPR / 2, STEPS = KS * KS * KSTEPS;
    half8 af[2][NR], bf[2][MR];
    const unsigned st_a = (unsigned)(slot * Tile::STAGE_BYTES);   // LDS byte address of the stage (smem starts at 0)
#define GTX_LOAD_FRAGS(STEP, BUF)                                                                              \
  {                                                                                                            \
    constexpr int tap__ = (STEP) / KSTEPS, ks__ = (STEP) % KSTEPS, ky__ = tap__ / KS, kx__ = tap__ % KS;       \
    _Pragma("unroll") for (int m = 0; m < MR; ++m) {                                                           \
      const int b = wave * MR + m;                                                                             \
      const int brow = b / TWB, bcol = b % TWB;                                                                \
      const int colbase = STRIDE == 1 ? bcol * 32 + kx__ : (kx__ & 1) * Tile::PWE + bcol * 32 + (kx__ >> 1);   \
      const int r = (brow * STRIDE + ky__) * PW + colbase + prow;                                              \
      const unsigned addr__ = st_a + r * RB + (((2 * ks__ + h) ^ Tile::swz(r)) << 4);                          \
      asm volatile("ds_read_b128 %0, %1" : "=v"(bf[BUF][m]) : "v"(addr__));                                    \
    }                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < NR; ++j) {                                                           \
      const unsigned addr__ = st_a + a_off[ks__];                                                              \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[BUF][j]) : "v"(addr__), "n"((tap__ * BN + 32 * j) * RB)); \
    }                                                                                                          \
  }
    GTX_LOAD_FRAGS(0, 0)
    static_for<STEPS>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int cb = s & 1;
      if constexpr (s + 1 < STEPS) {
        GTX_LOAD_FRAGS(s + 1, (s + 1) & 1)
        wait_frags<MR + NR, MR, NR>(af[cb], bf[cb]);
      } else {
        wait_frags<0, MR, NR>(af[cb], bf[cb]);
      }
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int m = 0; m < MR; ++m)
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb][j], bf[cb][m], acc[m][j], 0, 0, 0);
    });
#undef GTX_LOAD_FRAGS
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
#undef GTX_ISSUE_STAGE

  // ---- epilogue: bias + SiLU (+ residual) -> NHWC store, 4 consecutive channels per lane ----
  const float* __restrict__ bias = P.bias ? P.bias + ct * BN : nullptr;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int b = wave * MR + m;
    const int oy = oy0 + b / TWB, ox = ox0 + (b % TWB) * 32 + prow;
    if (oy >= outH || ox >= outW) continue;
    const size_t pix = ((size_t)n * outH + oy) * outW + ox;
    T* __restrict__ out = static_cast<T*>(P.out) + pix * P.out_cstride + P.out_coff + ct * BN;
    const T* __restrict__ res = P.res ? static_cast<const T*>(P.res) + pix * P.res_cstride + P.res_coff + ct * BN : nullptr;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int cl = 32 * j + 8 * g4 + 4 * h;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[m][j][4 * g4 + i] + (bias ? bias[cl + i] : 0.f);
          if (P.act) v[i] = silu2(v[i]);
        }
        if (res) {
          const half4 rv = *reinterpret_cast<const half4*>(res + cl);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += (float)rv[i];
        }
        half4 o;
        o[0] = (T)v[0]; o[1] = (T)v[1]; o[2] = (T)v[2]; o[3] = (T)v[3];
        *reinterpret_cast<half4*>(out + cl) = o;
      }
    }
  }
}

template <int KS, int STRIDE, int TH, int TWB, int NR, int CPR, int NS>
void launch2_t(const ConvGroup& g, hipStream_t stream) {
  using Tile = Tile2<KS, STRIDE, TH, TWB, NR, CPR, NS>;
  auto kern = conv_igemm2_kernel<KS, STRIDE, TH, TWB, NR, CPR, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(g.total_blocks), dim3(256), Tile::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace

// ---- v2 tile table -----------------------------------------------------------------------------

void conv2_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t s) {
  const int nr = c.bn / 32, cpr = c.kc / 8;
#define GTX_CASE2(KS, ST, TH, TWB, NRV, CPRV, NSV) \
  if (c.ks == KS && c.stride == ST && c.th == TH && c.tw == 32 * TWB && nr == NRV && cpr == CPRV && c.ns == NSV) return launch2_t<KS, ST, TH, TWB, NRV, CPRV, NSV>(g, s);
  GTX_CASE2(3, 1, 8, 1, 2, 2, 2) GTX_CASE2(3, 1, 8, 1, 2, 2, 3) GTX_CASE2(3, 1, 8, 1, 2, 2, 4)
  GTX_CASE2(3, 1, 8, 1, 1, 2, 2) GTX_CASE2(3, 1, 8, 1, 1, 2, 3) GTX_CASE2(3, 1, 8, 1, 1, 2, 4)
  GTX_CASE2(3, 1, 8, 1, 2, 4, 2)
  GTX_CASE2(1, 1, 1, 8, 2, 4, 2) GTX_CASE2(1, 1, 1, 8, 2, 4, 3) GTX_CASE2(1, 1, 1, 8, 2, 8, 2)
  GTX_CASE2(1, 1, 1, 8, 1, 4, 2) GTX_CASE2(1, 1, 1, 8, 1, 4, 3)
  GTX_CASE2(3, 2, 4, 1, 2, 2, 2) GTX_CASE2(3, 2, 4, 1, 2, 2, 3) GTX_CASE2(3, 2, 4, 1, 1, 2, 2)
  GTX_CASE2(3, 2, 8, 1, 2, 2, 2)
#undef GTX_CASE2
  fail(-3, "conv v2: no kernel for ks=%d stride=%d bn=%d kc=%d ns=%d", c.ks, c.stride, c.bn, c.kc, c.ns);
}

}  // namespace gtx
