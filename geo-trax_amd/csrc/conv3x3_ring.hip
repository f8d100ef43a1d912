// 3x3 stride-1 split-f16x3 convolution as a PERSISTENT workgroup per CU fed by LDS-DMA ("ring kernel"). gfx950 only.
//
// Same arithmetic, tile and LDS image as conv_igemm_split_kernel<3, 1, WN, 2, 1> (pair-format activations, three fp16 MFMAs
// per product, 8 x 16 output pixels x 32*WN couts per 4-wave workgroup, 16 input channels per K chunk); what changes is how
// a workgroup is fed and how long it lives:
//   * The round-2/3 kernel stages a chunk through registers (global -> VGPR -> ds_write) between two barriers and launches
//     one workgroup per tile. Measured: a workgroup alone on its CU needs ~3 800 cycles per chunk for 1 728 cycles of MFMA
//     issue (256 -> 256 at 60 x 60: 256 workgroups, 16 chunks, 35.6 us), three co-resident workgroups reach 75 % of the
//     matrix pipe inside their K loops, and most launches of the network hold fewer workgroups than the chip has slots.
//   * Here one workgroup per CU walks a contiguous range of tiles. A chunk's patch and weight taps arrive by
//     `buffer_load_dwordx4 ... lds` (no VGPR staging, no ds_write pass, out-of-image pixels zero-filled by the buffer range
//     check) into the other of two LDS stages while the MFMAs work on this one; the next tile's first chunk is requested
//     during this tile's last one, so the load latency of a tile start and the store tail of an epilogue are covered too.
//     One `s_barrier` per chunk; DMA retired by a counted `s_waitcnt vmcnt(N)` in front of it (N = the epilogue stores
//     that may still be in flight), never by __syncthreads() (which drains vmcnt to 0).
//   LDS per workgroup: 2 stages x (12 KB patch + 36 KB weights) + bias slot + epilogue staging = 131 KB -> one per CU.
// The LDS image is lane-linear per 1-KB DMA piece; the XOR swizzle the fragment reads expect is applied to the SOURCE
// address (cdna_hip_programming.md rule 21).
#include <hip/hip_runtime.h>

#include <mutex>

#include "conv_igemm.hpp"

// Diagnostic builds (`make stamp`) force-include csrc/diag/conv_split_diag.hpp, which defines these two hooks as clock
// stamps around a tile's K loop; the shipped object has none.
#ifndef GTXS_DIAG_LOOP_BEGIN
#define GTXS_DIAG_LOOP_BEGIN()
#define GTXS_DIAG_LOOP_END()
#endif

namespace gtx {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int WN>
struct RingTile {
  static constexpr int TH = 8, TW = 16, BN = 32 * WN, PH = 10, PW = 18, NPIX = PH * PW;   // 180 patch rows
  static constexpr int RB = 64;                                  // bytes per LDS row: hi0 hi1 lo0 lo1 (16 channels)
  static constexpr int PATCH_PIECES = 12;                        // 720 16-B slots -> 11.25 KB, padded to whole 1-KB pieces
  static constexpr int W_BYTES = 9 * BN * RB;                    // one (cout tile, chunk) block of the packed weights
  static constexpr int W_PIECES = (W_BYTES + 4095) / 4096 * 4;   // a multiple of 4: every wave issues the same count
  static constexpr int PATCH_PER_WAVE = PATCH_PIECES / 4, W_PER_WAVE = W_PIECES / 4;
  static constexpr int STAGE_BYTES = (PATCH_PIECES + W_PIECES) * 1024;
  static constexpr int EPI_OFF = 2 * STAGE_BYTES;
  static constexpr int EPI_PITCH = BN * 4 + 16;
  static constexpr int LDS_BYTES = EPI_OFF + 4 * 32 * EPI_PITCH;
  static constexpr int STORES_PER_WAVE = 32 * BN * 4 / 1024;     // epilogue store instructions per wave and tile
  static __host__ __device__ constexpr int swz(int row) { return (row >> 2) & 3; }
};

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

__device__ __forceinline__ void split4(const float (&v)[4], uint2& hi, uint2& lo, bool& sat) {
  half4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = __builtin_amdgcn_fmed3f(v[i], -65504.f, 65504.f);
    sat |= x != v[i];
    const _Float16 hh = (_Float16)x;
    h[i] = hh;
    l[i] = (_Float16)(x - (float)hh);
  }
  hi = *reinterpret_cast<const uint2*>(&h);
  lo = *reinterpret_cast<const uint2*>(&l);
}

typedef int int4v __attribute__((ext_vector_type(4)));

// A raw buffer descriptor over [base, base + bytes): accesses past the end return zeros / are dropped. Wave-uniform inputs.
__device__ __forceinline__ int4v make_desc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  int4v d;
  d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)) & 0xffff;     // stride 0: raw buffer
  d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000;
  return d;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// One LDS-DMA piece: 64 lanes x 16 B from desc[voff + soff] to LDS bytes [lds_dst, lds_dst + 1024), lane-linear.
// Inline asm on purpose: hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read that follows a DMA it can
// see (the builtin form), which would drain the prefetch of the NEXT chunk before this chunk's matrix phase starts. An asm
// load is absent from the compiler's bookkeeping; its completion is counted by hand (the vmcnt waits in the K loop).
// M0 carries the LDS destination and is restored (cdna_hip_programming.md, inline asm, M0).
__device__ __forceinline__ void dma16(const int4v& desc_in, unsigned lds_dst_in, unsigned voff, unsigned soff_in) {
  // the scalar operands are wave-uniform by construction; readfirstlane makes that provable (a no-op where it already is)
  int4v desc;
#pragma unroll
  for (int i = 0; i < 4; ++i) desc[i] = __builtin_amdgcn_readfirstlane(desc_in[i]);
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds_dst_in), soff = __builtin_amdgcn_readfirstlane(soff_in);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_dst), "v"(voff), "s"(desc), "s"(soff)
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(lds_void*)p; }

constexpr unsigned kOob = 0x80000000u;     // a byte offset beyond every buffer of the network: the range check returns 0

// SINGLE: the launch has one member (every convolution but the Detect head's grouped stages): g.p[0] is a compile-time
// index, so its fields are loop-invariant scalar loads instead of a dozen dependent ones per tile.
template <int WN, bool SINGLE>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 1)))
void conv3x3_ring_kernel(const ConvGroup g) {
  using T = RingTile<WN>;
  constexpr int BN = T::BN, PW = T::PW, RB = T::RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // this workgroup's tiles: an equal share of its XCD's range of logical blocks (ConvGroup::xcd_begin), contiguous
  const int xcd = blockIdx.x & 7, idx = (int)(blockIdx.x >> 3), per_xcd = (int)(gridDim.x >> 3);
  const int xb = g.xcd_begin[xcd], xlen = g.xcd_begin[xcd + 1] - xb;
  const int t_begin = xb + (int)((long)idx * xlen / per_xcd), t_end = xb + (int)((long)(idx + 1) * xlen / per_xcd);
  if (t_begin >= t_end) return;

  const int prow = lane & 31, h = lane >> 5;
  const int tcol = prow & 15;
  const int trow0 = 2 * wave + (prow >> 4);
  const int p0 = trow0 * PW + tcol;

  // ---- per-tile state (wave-uniform unless noted) ----
  struct TileCtx {
    int pi, ct, n, oy0, ox0, nch;
    int4v din, dw;                      // buffer descriptors of the member's input tensor and packed weights
    unsigned w0;                        // byte offset of this wave's first weight piece of chunk 0
    unsigned pv[T::PATCH_PER_WAVE];     // per lane: byte offset of this lane's 16 B of each of the wave's patch pieces (chunk 0)
  };
  auto decode = [&](int L, TileCtx& c) {
    int pi = 0;
    if (!SINGLE) {
#pragma unroll
      for (int i = 1; i < kMaxGroup; ++i)
        if (i < g.count && L >= g.p[i].block_begin) pi = i;
    }
    const ConvProblem& P = g.p[SINGLE ? 0 : pi];
    const int lb = L - P.block_begin;
    c.pi = pi;
    c.ct = lb % P.n_ct;
    const int pt = lb / P.n_ct;
    const int tx = pt % P.tiles_x, t2 = pt / P.tiles_x;
    c.n = t2 / P.tiles_y;
    c.oy0 = (t2 % P.tiles_y) * T::TH;
    c.ox0 = tx * T::TW;
    c.nch = P.Cin >> 4;
    c.din = make_desc(P.in, (unsigned)P.N * P.H * P.W * P.in_cstride * 4u);
    c.dw = make_desc(P.wpack, (unsigned)P.n_ct * (unsigned)c.nch * T::W_BYTES);
    c.w0 = __builtin_amdgcn_readfirstlane((unsigned)(c.ct * c.nch) * T::W_BYTES + wave * 1024);
    const int iy0 = c.oy0 - 1, ix0 = c.ox0 - 1;
#pragma unroll
    for (int i = 0; i < T::PATCH_PER_WAVE; ++i) {
      const int s = (wave + 4 * i) * 64 + lane;          // 16-B slot of the patch image
      const int p = s >> 2, cs = s & 3;
      const int ch = cs ^ T::swz(p);                      // logical chunk held by that slot: hi0 hi1 lo0 lo1
      const int py = p / PW, px = p - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool inb = p < T::NPIX && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
      c.pv[i] = inb ? (unsigned)((((c.n * P.H + iy) * P.W + ix) * P.in_cstride + P.in_coff) * 4 + (ch & 1) * 32 + (ch >> 1) * 16) : kOob;
    }
  };
  // A request = the DMA pieces of one (tile, chunk) into one stage: PATCH_PER_WAVE + W_PER_WAVE pieces per wave (+ the bias
  // row, wave 0, chunk 0). It is issued piece by piece (request_piece) so that the K loop can spread the pieces over the taps
  // of the running matrix phase instead of issuing 12 of them back to back in front of it.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  struct Request {
    int4v din, dw;
    unsigned stage, wbase, csoff;          // LDS byte address of this wave's first piece; weight block offset; channel offset of the chunk
    unsigned pv[T::PATCH_PER_WAVE];        // per lane
  };
  auto make_request = [&](const TileCtx& c, int chunk, int st, Request& r) {
    r.stage = lds0 + st * T::STAGE_BYTES + wave * 1024;
    r.din = c.din;
    r.dw = c.dw;
    r.wbase = c.w0 + (unsigned)chunk * T::W_BYTES;
    r.csoff = (unsigned)chunk * 64;
#pragma unroll
    for (int i = 0; i < T::PATCH_PER_WAVE; ++i) r.pv[i] = c.pv[i];
  };
  // piece k of a request: 0 .. PATCH_PER_WAVE-1 patch, then W_PER_WAVE weight pieces
  auto request_piece = [&](const Request& r, int k) {
    if (k < T::PATCH_PER_WAVE) {
      dma16(r.din, r.stage + k * 4096, r.pv[k], r.csoff);
    } else {
      const int i = k - T::PATCH_PER_WAVE;
      // pieces past the block (BN = 32: 18 real pieces of 20) carry an out-of-range offset and land as zeros in the stage's padding
      const bool real = (4 * i + 3) * 1024 < T::W_BYTES || (wave + 4 * i) * 1024 < T::W_BYTES;
      dma16(r.dw, r.stage + (T::PATCH_PIECES + 4 * i) * 1024, real ? (unsigned)(lane * 16) : kOob, r.wbase + i * 4096);
    }
  };
  constexpr int kPieces = T::PATCH_PER_WAVE + T::W_PER_WAVE;

  TileCtx cur, nxt;
  decode(t_begin, cur);
  {
    Request r0;
    make_request(cur, 0, 0, r0);
#pragma unroll
    for (int k = 0; k < kPieces; ++k) request_piece(r0, k);
  }
  int st = 0;
  bool sat = false;
  bool stores_pending = false;

  for (int L = t_begin; L < t_end; ++L) {
    const ConvProblem& P = g.p[SINGLE ? 0 : cur.pi];
    floatx16 acc[WN];
    uintx4 bias_r[WN][4];
    const bool more_tiles = L + 1 < t_end;
    GTXS_DIAG_LOOP_BEGIN()
    for (int chunk = 0; chunk < cur.nch; ++chunk) {
      // this wave's pieces of (tile, chunk) have landed; only the previous tile's epilogue stores may be younger
      if (chunk == 0 && stores_pending) {
        if (WN == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");   // everyone's pieces landed; everyone is done reading the other stage
      // what to request while this chunk computes: the tile's next chunk, or the next tile's first one
      Request rq;
      bool have_rq = true;
      if (chunk + 1 < cur.nch) {
        make_request(cur, chunk + 1, st ^ 1, rq);
      } else if (more_tiles) {
        decode(L + 1, nxt);
        make_request(nxt, 0, st ^ 1, rq);
      } else {
        have_rq = false;
      }
      const char* lds_patch = smem + st * T::STAGE_BYTES;
      const char* lds_w = lds_patch + T::PATCH_PIECES * 1024;
      if (chunk == 0) {
        // The tile's bias values go to registers now (ordinary range-checked loads, far ahead of the epilogue that adds them: a
        // null bias pointer makes a zero-length buffer, i.e. zeros) and the accumulators start at zero.
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(P.bias, P.bias ? (unsigned)P.n_ct * BN * 4u : 0u);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            bias_r[j][g4] = __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((cur.ct * BN + 32 * j + 8 * g4 + 4 * h) * 4), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][4 * g4 + i] = 0.f;
          }
      }
      // ---- matrix phase: 9 taps x (WN cout tiles x 3 MFMAs). One wave per SIMD: nothing else fills the matrix pipe while this
      // wave waits, so the fragment reads run TWO taps ahead of the MFMAs that use them (three register sets; with one tap of
      // lead a read had 2-4 MFMAs = 64-128 cycles to return and the phase ran at ~55 cycles per MFMA), and the DMA pieces of
      // the next request are issued one or two per tap, between the MFMAs, instead of in a burst in front of the phase.
      half8 bh[3], bl[3], ah[3][WN], al[3][WN];
#define GTXR_LOAD_FRAGS(TAP, SLOT)                                                                        \
      {                                                                                                     \
        const int p__ = p0 + ((TAP) / 3) * PW + ((TAP) % 3);                                                \
        const char* pr__ = lds_patch + p__ * RB;                                                            \
        bh[SLOT] = *reinterpret_cast<const half8*>(pr__ + ((h ^ T::swz(p__)) << 4));                        \
        bl[SLOT] = *reinterpret_cast<const half8*>(pr__ + (((2 + h) ^ T::swz(p__)) << 4));                  \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                                    \
          const int nrow__ = 32 * j + prow;                                                                 \
          const char* wr__ = lds_w + ((TAP) * BN + nrow__) * RB;                                            \
          ah[SLOT][j] = *reinterpret_cast<const half8*>(wr__ + ((h ^ T::swz(nrow__)) << 4));                \
          al[SLOT][j] = *reinterpret_cast<const half8*>(wr__ + (((2 + h) ^ T::swz(nrow__)) << 4));          \
        }                                                                                                   \
      }
      GTXR_LOAD_FRAGS(0, 0)
      GTXR_LOAD_FRAGS(1, 1)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        __builtin_amdgcn_sched_barrier(0);
        if (have_rq) {
#pragma unroll
          for (int k = 0; k < kPieces; ++k)
            if (k * 9 / kPieces == tap) request_piece(rq, k);
        }
        if (tap + 2 < 9) {
          if ((tap + 2) % 3 == 0) GTXR_LOAD_FRAGS(tap + 2, 0)
          else if ((tap + 2) % 3 == 1) GTXR_LOAD_FRAGS(tap + 2, 1)
          else GTXR_LOAD_FRAGS(tap + 2, 2)
        }
        // small terms first; the cout tiles alternate so that two MFMAs on one accumulator are never adjacent
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tap % 3][j], bh[tap % 3], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % 3][j], bl[tap % 3], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tap % 3][j], bh[tap % 3], acc[j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 3 * WN; ++i) {                     // one fragment read (of the tap after next) behind every MFMA of this one
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef GTXR_LOAD_FRAGS
      st ^= 1;
    }
    GTXS_DIAG_LOOP_END()

    // ---- epilogue (as conv_igemm_split.hip): scale, SiLU, residual, split, LDS transpose in a wave-private area, whole-line stores ----
    {
      const float sc = P.acc_scale;
      const int cvalid = P.Cout - cur.ct * BN;
      const bool plain = P.out_plain != 0;
      const int oy = cur.oy0 + trow0, ox = cur.ox0 + tcol;
      constexpr int PITCH = T::EPI_PITCH;
      char* stg = smem + T::EPI_OFF + wave * (32 * PITCH);
      const bool inside = oy < P.Ho && ox < P.Wo;
      const size_t pix = inside ? ((size_t)cur.n * P.Ho + oy) * P.Wo + ox : 0;
      // residual: every lane loads its 16-byte chunk of every group up front (range-checked buffer loads, no branches: one
      // wait for all of them instead of a round trip per group)
      uintx4 rc[WN][4];
      if (P.res) {
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(P.res, (unsigned)P.N * P.Ho * P.Wo * P.res_cstride * 4u);
        const unsigned rbase = (unsigned)((pix * P.res_cstride + P.res_coff + cur.ct * BN) * 4);
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int cl = 32 * j + 8 * g4 + 4 * h;
            rc[j][g4] = __builtin_amdgcn_raw_buffer_load_b128(rres, (inside && cl < cvalid) ? rbase + cl * 4 : kOob, 0, 0);
          }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int cl = 32 * j + 8 * g4 + 4 * h;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = fmaf(acc[j][4 * g4 + i], sc, __uint_as_float(bias_r[j][g4][i]));
            if (P.act) v[i] = silu_f(v[i]);
          }
          if (P.res) {
            const auto sx = __builtin_amdgcn_permlane32_swap(rc[j][g4][0], rc[j][g4][2], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(rc[j][g4][1], rc[j][g4][3], false, false);
            const unsigned hw[2] = {sx[0], sy[0]}, lw[2] = {sx[1], sy[1]};
            const half4 rh = *reinterpret_cast<const half4*>(hw), rl = *reinterpret_cast<const half4*>(lw);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rh[i] + (float)rl[i];
          }
          if (plain) {
            *reinterpret_cast<float4*>(stg + prow * PITCH + cl * 4) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 hi, lo;
            split4(v, hi, lo, sat);
            const auto sx = __builtin_amdgcn_permlane32_swap(hi.x, lo.x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(hi.y, lo.y, false, false);
            *reinterpret_cast<uint4*>(stg + prow * PITCH + cl * 4) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      }
      constexpr int LPP = BN / 4, PPI = 64 / LPP;
      // Buffer stores: every wave executes exactly STORES_PER_WAVE store instructions per tile (the counted vmcnt above relies
      // on it); lanes outside the image or past Cout carry a range-checked offset and are dropped by the hardware.
      const __amdgpu_buffer_rsrc_t rout = make_rsrc(P.out, (unsigned)P.N * P.Ho * P.Wo * P.out_cstride * 4u);
#pragma unroll
      for (int it = 0; it < 32 / PPI; ++it) {
        const int p = it * PPI + lane / LPP, q = lane % LPP;
        const int py = cur.oy0 + 2 * wave + (p >> 4), px = cur.ox0 + (p & 15);
        const uint4 val = *reinterpret_cast<const uint4*>(stg + p * PITCH + q * 16);
        const bool ok = py < P.Ho && px < P.Wo && (q >> 1) * 8 < cvalid;
        const unsigned off = ok ? (unsigned)((((cur.n * P.Ho + py) * P.Wo + px) * P.out_cstride + P.out_coff + cur.ct * BN + q * 4) * 4) : kOob;
        __builtin_amdgcn_raw_buffer_store_b128(uintx4{val.x, val.y, val.z, val.w}, rout, off, 0, 0);
      }
      static_assert(32 / PPI == T::STORES_PER_WAVE, "store count");
      stores_pending = true;
    }
    cur = nxt;
  }
  // sat_flag of the launch's first member (the detector points every member at the same word)
  if (g.p[0].sat_flag && __builtin_amdgcn_ballot_w64(sat) != 0 && lane == 0) atomicOr(g.p[0].sat_flag, 1);
}

template <int WN, bool SINGLE>
void launch_ring_t(const ConvGroup& g, hipStream_t stream) {
  using T = RingTile<WN>;
  auto kern = conv3x3_ring_kernel<WN, SINGLE>;
  static std::once_flag once;
  std::call_once(once, [&] {
    GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  });
  const int grid = std::min(256, (g.total_blocks + 7) / 8 * 8);     // one workgroup per CU; fewer when the launch has fewer tiles
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), T::LDS_BYTES, stream, g);
  GTX_HIP(hipGetLastError());
}

}  // namespace

void conv_ring_launch(const ConvGroup& g, const ConvConfig& c, hipStream_t s) {
  GTX_CHECK(c.ks == 3 && c.stride == 1 && c.kc == 16 && c.th == 8 && (c.bn == 32 || c.bn == 64), "conv (ring): 3x3 stride 1, 16-channel chunks only");
  for (int i = 0; i < g.count; ++i) GTX_CHECK(g.p[i].Cin >= 32, "conv (ring): needs at least two K chunks (Cin %d)", g.p[i].Cin);
  if (c.bn == 64) return g.count == 1 ? launch_ring_t<2, true>(g, s) : launch_ring_t<2, false>(g, s);
  return g.count == 1 ? launch_ring_t<1, true>(g, s) : launch_ring_t<1, false>(g, s);
}

}  // namespace gtx
