// Brute-force L2 2-nearest-neighbour matching of 128-d descriptors on MFMA (gfx950).
// Replaces cv2.BFMatcher(NORM_L2).knnMatch(query, train, k=2) as stabilo runs it for the
// orthophoto registration (reference: geotrax/utils/registration.py:59-85, matcher_name='bf',
// up to 250 000 x 250 000 x 128 -- the one big dense contraction of the product).
//
// Descriptors are unit-L2-norm (RootSIFT), so the nearest neighbour is the largest dot product.
// The search runs in fp16 (fp32 accumulate); the two winners of every query are then re-measured
// in fp32 from the fp32 descriptors, so the distances handed to the ratio test are exact.
//
// Work decomposition
//   workgroup (4 waves)  -> 128 queries x one split of the train set
//   wave                 -> 32 queries: their 128-d rows live in registers as 8 MFMA B fragments
//   train tiles          -> 128 rows x 256 B staged in LDS (XOR-swizzled, double buffered, register
//                           prefetch), shared by the 4 waves; weights-as-A orientation: D[train][query],
//                           so a lane's 16 accumulators are 16 train rows of ONE query and the running
//                           top-2 is lane-local (no cross-lane traffic until the final h=0/1 merge)
//   top-2 update         -> max of the 16 dots (v_max3 tree); the per-element insertion runs only when
//                           that max beats the current second best, which after the first tiles is rare
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "match_l2.hpp"

namespace gtx {

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kD = 128;            // descriptor length
constexpr int kTile = 128;         // train rows per LDS tile
constexpr int kRowB = kD * 2;      // bytes per fp16 row

struct Top2 {
  float b1, b2;
  int i1, i2;
};

__device__ __forceinline__ void top2_insert(Top2& t, float v, int idx) {
  if (v > t.b1) { t.b2 = t.b1; t.i2 = t.i1; t.b1 = v; t.i1 = idx; }
  else if (v > t.b2) { t.b2 = v; t.i2 = idx; }
}

__global__ __launch_bounds__(256) void match2nn_kernel(const _Float16* __restrict__ q, int nq, const _Float16* __restrict__ t, int nt,
                                                        int rows_per_split, Top2* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) char s_t[2][kTile * kRowB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = min(q0 + col, nq - 1);

  // query fragments: k-step ks covers channels 16 ks .. 16 ks + 15, this lane holds 8 of them
  half8 bq[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) bq[ks] = *reinterpret_cast<const half8*>(q + (size_t)qi * kD + 16 * ks + 8 * h);

  const int r_begin = blockIdx.y * rows_per_split;
  const int r_end = min(r_begin + rows_per_split, nt);
  const int ntiles = (r_end - r_begin + kTile - 1) / kTile;

  // staging: thread handles 8 of the 2048 16-B chunks of a tile
  uint4 pre[8];
#define GTX_MATCH_PREFETCH(TI)                                                        \
  {                                                                                   \
    const int rb__ = r_begin + (TI) * kTile;                                          \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                   \
      const int c__ = tid + 256 * s, row__ = c__ >> 4, ch__ = c__ & 15;               \
      uint4 v__ = make_uint4(0, 0, 0, 0);                                             \
      if (rb__ + row__ < r_end) v__ = *reinterpret_cast<const uint4*>(t + (size_t)(rb__ + row__) * kD + ch__ * 8); \
      pre[s] = v__;                                                                   \
    }                                                                                 \
  }
#define GTX_MATCH_COMMIT(BUF)                                                         \
  {                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                   \
      const int c__ = tid + 256 * s, row__ = c__ >> 4, ch__ = c__ & 15;               \
      *reinterpret_cast<uint4*>(s_t[BUF] + row__ * kRowB + ((ch__ ^ (row__ & 15)) << 4)) = pre[s]; \
    }                                                                                 \
  }

  Top2 best{-3.0e38f, -3.0e38f, -1, -1};
  if (ntiles > 0) GTX_MATCH_PREFETCH(0)
  for (int ti = 0; ti < ntiles; ++ti) {
    const int buf = ti & 1;
    GTX_MATCH_COMMIT(buf)
    __syncthreads();                       // tile ti visible; everyone is past tile ti-1 (buffer buf^1 is free next round)
    if (ti + 1 < ntiles) GTX_MATCH_PREFETCH(ti + 1)
    const char* st = s_t[buf];
    const int tile_row0 = r_begin + ti * kTile;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      floatx16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int row = 32 * blk + col;     // A fragment row of this lane
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const half8 a = *reinterpret_cast<const half8*>(st + row * kRowB + (((2 * ks + h) ^ (row & 15)) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[ks], acc, 0, 0, 0);
      }
      float m = acc[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) m = fmaxf(m, acc[i]);
      if (m > best.b2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int idx = tile_row0 + 32 * blk + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (idx < r_end) top2_insert(best, acc[i], idx);
        }
      }
    }
  }
#undef GTX_MATCH_PREFETCH
#undef GTX_MATCH_COMMIT
  // merge the two half-waves that hold the same query (h = 0 / 1)
  Top2 o;
  o.b1 = __shfl_xor(best.b1, 32); o.b2 = __shfl_xor(best.b2, 32);
  o.i1 = __shfl_xor(best.i1, 32); o.i2 = __shfl_xor(best.i2, 32);
  if (o.i1 >= 0) top2_insert(best, o.b1, o.i1);
  if (o.i2 >= 0) top2_insert(best, o.b2, o.i2);
  if (h == 0 && q0 + col < nq) part[(size_t)blockIdx.y * nq + q0 + col] = best;
}

// Merge of the per-split partials, exact fp32 distances of the two winners (ordered by the exact
// distance), one wave per query.
__global__ __launch_bounds__(256) void match2nn_finish_kernel(const Top2* __restrict__ part, int nsplit, const float* __restrict__ qf, int nq,
                                                               const float* __restrict__ tf, int nt, int* __restrict__ idx1,
                                                               int* __restrict__ idx2, float* __restrict__ d1, float* __restrict__ d2) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  Top2 b{-3.0e38f, -3.0e38f, -1, -1};
  for (int s = 0; s < nsplit; ++s) {
    const Top2 p = part[(size_t)s * nq + qi];
    if (p.i1 >= 0) top2_insert(b, p.b1, p.i1);
    if (p.i2 >= 0) top2_insert(b, p.b2, p.i2);
  }
  float e1 = 0.f, e2 = 0.f;
  const float qa = qf[(size_t)qi * kD + lane], qb = qf[(size_t)qi * kD + 64 + lane];
  if (b.i1 >= 0) {
    const float x = qa - tf[(size_t)b.i1 * kD + lane], y = qb - tf[(size_t)b.i1 * kD + 64 + lane];
    e1 = x * x + y * y;
  }
  if (b.i2 >= 0) {
    const float x = qa - tf[(size_t)b.i2 * kD + lane], y = qb - tf[(size_t)b.i2 * kD + 64 + lane];
    e2 = x * x + y * y;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { e1 += __shfl_xor(e1, o); e2 += __shfl_xor(e2, o); }
  if (lane == 0) {
    float f1 = b.i1 >= 0 ? sqrtf(e1) : 3.0e38f, f2 = b.i2 >= 0 ? sqrtf(e2) : 3.0e38f;
    int j1 = b.i1, j2 = b.i2;
    if (f2 < f1 || (f2 == f1 && j2 >= 0 && j2 < j1)) { const float tf_ = f1; f1 = f2; f2 = tf_; const int tj = j1; j1 = j2; j2 = tj; }
    idx1[qi] = j1; idx2[qi] = j2; d1[qi] = f1; d2[qi] = f2;
  }
}

__global__ void to_half_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (_Float16)src[i];
}

}  // namespace

int match2nn_splits(int nq, int nt) {
  // enough workgroups to fill 256 CUs a few times over, but splits of at least 8 tiles
  const int qblocks = cdiv(nq, 128);
  int s = cdiv(2048, std::max(qblocks, 1));
  s = std::min(s, std::max(1, nt / (8 * kTile)));
  return std::max(1, std::min(s, 64));
}

size_t match2nn_workspace_bytes(int nq, int nt) { return (size_t)match2nn_splits(nq, nt) * nq * sizeof(Top2); }

void descriptors_to_half(const float* src, void* dst, size_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(to_half_kernel, dim3((unsigned)cdiv((long)n, 256L)), dim3(256), 0, s, src, static_cast<_Float16*>(dst), n);
  GTX_HIP(hipGetLastError());
}

void match2nn(const void* q_f16, const float* q_f32, int nq, const void* t_f16, const float* t_f32, int nt, void* workspace,
              int* idx1, int* idx2, float* d1, float* d2, hipStream_t s) {
  if (nq == 0) return;
  const int nsplit = match2nn_splits(nq, nt);
  int rows = cdiv(std::max(nt, 1), nsplit);
  rows = cdiv(rows, kTile) * kTile;
  hipLaunchKernelGGL(match2nn_kernel, dim3(cdiv(nq, 128), nsplit), dim3(256), 0, s, static_cast<const _Float16*>(q_f16), nq,
                     static_cast<const _Float16*>(t_f16), nt, rows, static_cast<Top2*>(workspace));
  hipLaunchKernelGGL(match2nn_finish_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, s, static_cast<const Top2*>(workspace), nsplit, q_f32,
                     nq, t_f32, nt, idx1, idx2, d1, d2);
  GTX_HIP(hipGetLastError());
}

double match2nn_flops(int nq, int nt) { return 2.0 * nq * (double)nt * kD; }

}  // namespace gtx
