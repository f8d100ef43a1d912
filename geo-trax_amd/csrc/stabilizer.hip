// GPU homography stabilizer for gfx950: ORB-style keypoints (integer-exact pyramid, FAST-9/16,
// Harris ranking, intensity-centroid orientation, steered BRIEF), brute-force Hamming 2-NN with
// Lowe's ratio test, and a massively parallel RANSAC homography with an f64 refit on the host.
//
// Stands in for stabilo.Stabilizer as the reference drives it (geotrax/extract.py:139,177-187;
// parameters geotrax/cfg/default.yaml:100-145): detector 'orb', matcher 'bf', filter 'ratio',
// transformation 'projective', downsample_ratio 0.5, foreground mask from the frame's boxes.
// stabilo delegates to OpenCV (ORB_create / BFMatcher.knnMatch / findHomography USAC_MAGSAC);
// none of that source is in the reference tree, so the stages are specified here and restated
// in oracle/stabilo_ref.py. Every stage up to and including matching is integer arithmetic and
// is bit-exact against the oracle; the homography is f64 and compared by reprojection distance.
//
// All image work is HBM/L2-bound byte traffic (~7 MB of pyramid per frame): coalesced row
// accesses, wave ballots / shuffles for reductions, LDS only for the per-keypoint patch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "detector.hpp"   // gtx_ctx
#include "geometry.hpp"
#include "stabilizer.hpp"

namespace gtx {

namespace {

constexpr int kPyrLevels = 8;
constexpr int kBorder = 31;        // ORB edgeThreshold
constexpr int kPatchR = 20;        // patch radius staged per keypoint: 17 (rotated pattern) + 3 (blur)
constexpr int kPatchW = 2 * kPatchR + 1;
constexpr int kAngleBins = 256;
constexpr int kMaxRects = 1024;

// FAST candidates of a level are appended to kCandSub sub-lists (tile index mod kCandSub), each with its own counter:
// one reservation per tile on a single counter per level serialised ~18 k same-address atomics per frame in L2
// (fast_detect 31 us with everything masked, 66-74 us with real masks).
constexpr int kCandSub = 16;
constexpr int kCounterInts = kPyrLevels * kCandSub + kPyrLevels + 256 * kPyrLevels;   // cand_n, elig_n, score histograms

struct Level {
  const uint8_t* img;   // this level's pixels (level 0 may live in a caller-owned buffer)
  int w, h;
  int off;          // byte offset of this level in the pyramid / score buffers
  int cand_off;     // first candidate slot of this level
  int cand_cap;     // kCandSub sub-lists of sub_cap slots each
  int sub_cap;
  int n_want;       // keypoints to keep
  int kp_off;       // first output keypoint slot
  float scale;      // level pixel -> level-0 pixel
  int tiles_x;      // 64x16-pixel tiles across
  int tile_begin;   // first flattened tile of this level
};

struct Levels {
  Level l[kPyrLevels];
  int n;
  int n_tiles;      // over all levels
  int tab_off[kPyrLevels];   // pyr_resize_kernel: first entry of level i's column table (its row table follows) in the table buffer
  // pyr_group_kernel: the resizes of levels 1.. in launches of consecutive levels (group g computes levels [grp_first[g],
  // grp_first[g] + grp_n[g])), and the LDS rectangle buffer each launch needs (largest rectangle of any of its tiles)
  int n_groups;
  int grp_first[kPyrLevels], grp_n[kPyrLevels], grp_buf[kPyrLevels];
};

// ------------------------------------------------------------------ gray / pyramid
__device__ __forceinline__ int bgr2gray_u8(int b, int g, int r) { return (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14; }

__global__ __launch_bounds__(256) void gray_kernel(const uint8_t* __restrict__ bgr, int h, int w, int half,
                                                   uint8_t* __restrict__ out, int oh, int ow) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  if (half) {
    const uint8_t* r0 = bgr + ((size_t)(2 * y) * w + 2 * x) * 3;
    const uint8_t* r1 = r0 + (size_t)w * 3;
    const int s = bgr2gray_u8(r0[0], r0[1], r0[2]) + bgr2gray_u8(r0[3], r0[4], r0[5]) +
                  bgr2gray_u8(r1[0], r1[1], r1[2]) + bgr2gray_u8(r1[3], r1[4], r1[5]);
    out[(size_t)y * ow + x] = (uint8_t)((s + 2) >> 2);
  } else {
    const uint8_t* p = bgr + ((size_t)y * w + x) * 3;
    out[(size_t)y * ow + x] = (uint8_t)bgr2gray_u8(p[0], p[1], p[2]);
  }
}

// Integer bilinear resize of one pyramid level from the one above it (16.16 source coordinates at pixel centres, 11-bit
// weights, round to nearest): bit-exact on any machine. The source column / row and weight of every destination column /
// row depend on the level sizes only, so they are tabulated once when the stabilizer is created (pyr_src; the table entry
// packs source index, weight and whether the second tap is a different pixel) instead of two 64-bit divisions per pixel.
// A thread produces 4 consecutive pixels of a row and stores them as one dword when the address allows.
// (A single-launch variant -- a workgroup per 64x16 tile of the last level rebuilding everything below it in LDS -- was
// built and measured: 185 us per frame in situ against 7 x 13.5 for the per-level launches it replaced; too little
// parallelism per launch and a dependent LDS chain per pixel. Removed.)
__host__ __device__ inline unsigned pyr_src(int x, int sw, int dw) {
  long fxp = ((long)(2 * x + 1) * sw * 32768) / dw - 32768;
  if (fxp < 0) fxp = 0;
  const unsigned x0 = (unsigned)(fxp >> 16), fx = (unsigned)((fxp >> 5) & 2047);
  return x0 | (fx << 16) | ((x0 + 1 < (unsigned)sw ? 1u : 0u) << 27);       // x0 <= sw - 1 always
}

__global__ __launch_bounds__(256) void pyr_resize_kernel(const uint8_t* __restrict__ src, int sw, uint8_t* __restrict__ dst, int dw,
                                                         const unsigned* __restrict__ tab_x, const unsigned* __restrict__ tab_y) {
  const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
  if (x >= dw) return;
  const unsigned ty = tab_y[y];
  const unsigned fy = (ty >> 16) & 2047;
  const uint8_t* r0 = src + (size_t)(ty & 0xffff) * sw;
  const uint8_t* r1 = r0 + ((ty >> 27) ? sw : 0);
  unsigned out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned tx = tab_x[min(x + k, dw - 1)];
    const unsigned xa = tx & 0xffff, xb = xa + (tx >> 27), fx = (tx >> 16) & 2047;
    const unsigned top = r0[xa] * (2048 - fx) + r0[xb] * fx;
    const unsigned bot = r1[xa] * (2048 - fx) + r1[xb] * fx;
    out |= ((top * (2048 - fy) + bot * fy + (1u << 21)) >> 22) << (8 * k);   // < 2^31: exact in 32 bits
  }
  uint8_t* d = dst + (size_t)y * dw + x;
  if (x + 4 <= dw && (reinterpret_cast<uintptr_t>(d) & 3) == 0) {
    *reinterpret_cast<unsigned*>(d) = out;
  } else {
    for (int k = 0; k < 4 && x + k < dw; ++k) d[k] = (uint8_t)(out >> (8 * k));
  }
}

// Several consecutive pyramid levels in ONE launch (round 4: the eight-level pyramid is two launches, levels 1-3 and 4-7,
// instead of seven). A workgroup owns a 64x16 tile of the group's LAST level; walking the resize tables backwards gives the
// rectangle of every level in front of it that the tile depends on, down to the group's source level. That source rectangle
// is staged in LDS, then level after level is computed from the rectangle before it -- the same integer bilinear expression
// as pyr_resize_kernel on the same operands, so every byte is the per-level launches' byte -- into the other LDS buffer and
// written to the level's image. Rectangles of neighbouring tiles overlap by the one or two pixels of bilinear support:
// those pixels are computed and stored by both workgroups, with identical values. A tile on the right / bottom edge of its
// level takes the rest of every level in front of it along, so that the levels are written completely.
// Unlike the whole pyramid as one launch (185 us, see above) a group's chain is 3-4 short stages and the first group still
// has 720 workgroups at 1920 x 1080.
constexpr int kPyrGroupMax = 4;
struct PyrGroup {
  int n;                            // levels computed by this launch
  const uint8_t* src;               // the level in front of the group
  int sw, sh;
  uint8_t* dst[kPyrGroupMax];
  int w[kPyrGroupMax], h[kPyrGroupMax];
  const unsigned* tx[kPyrGroupMax]; // column table of the level, its row table follows (pyr_src)
  int buf_bytes;                    // one LDS rectangle buffer (two are used)
};

__global__ __launch_bounds__(256) void pyr_group_kernel(const PyrGroup G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_pyr[];
  __shared__ int s_rect[kPyrGroupMax + 1][4];      // x0, y0, x1, y1 (inclusive); [0] = source level, [k + 1] = level k of the group
  const int tid = threadIdx.x;
  const int last = G.n - 1;
  if (tid == 0) {
    int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
    int x1 = min(x0 + 63, G.w[last] - 1), y1 = min(y0 + 15, G.h[last] - 1);
    for (int k = last; k >= 0; --k) {
      s_rect[k + 1][0] = x0; s_rect[k + 1][1] = y0; s_rect[k + 1][2] = x1; s_rect[k + 1][3] = y1;
      const unsigned* tabx = G.tx[k];
      const unsigned* taby = tabx + G.w[k];
      const int pw = k == 0 ? G.sw : G.w[k - 1], ph = k == 0 ? G.sh : G.h[k - 1];
      const bool right = x1 == G.w[k] - 1, bottom = y1 == G.h[k] - 1;
      const unsigned a = tabx[x0], b = tabx[x1], c = taby[y0], d = taby[y1];
      x0 = (int)(a & 0xffff); y0 = (int)(c & 0xffff);
      x1 = right ? pw - 1 : (int)((b & 0xffff) + (b >> 27));
      y1 = bottom ? ph - 1 : (int)((d & 0xffff) + (d >> 27));
    }
    s_rect[0][0] = x0; s_rect[0][1] = y0; s_rect[0][2] = x1; s_rect[0][3] = y1;
  }
  __syncthreads();
  uint8_t* bufs[2] = {s_pyr, s_pyr + G.buf_bytes};
  {
    const int rx = s_rect[0][0], ry = s_rect[0][1], rw = s_rect[0][2] - rx + 1, rh = s_rect[0][3] - ry + 1;
    for (int i = tid; i < rw * rh; i += 256) bufs[0][i] = G.src[(size_t)(ry + i / rw) * G.sw + rx + i % rw];
  }
  __syncthreads();
  for (int k = 0; k < G.n; ++k) {
    const uint8_t* in = bufs[k & 1];
    uint8_t* out = bufs[(k + 1) & 1];
    const int sx = s_rect[k][0], sy = s_rect[k][1], sw = s_rect[k][2] - sx + 1;
    const int dx = s_rect[k + 1][0], dy = s_rect[k + 1][1], dw = s_rect[k + 1][2] - dx + 1, dh = s_rect[k + 1][3] - dy + 1;
    const unsigned* tabx = G.tx[k];
    const unsigned* taby = tabx + G.w[k];
    uint8_t* img = G.dst[k];
    const int W = G.w[k];
    for (int i = tid; i < dw * dh; i += 256) {
      const int x = dx + i % dw, y = dy + i / dw;
      const unsigned tx = tabx[x], ty = taby[y];
      const unsigned fy = (ty >> 16) & 2047, fx = (tx >> 16) & 2047;
      const int o0 = ((int)(ty & 0xffff) - sy) * sw - sx, o1 = o0 + ((ty >> 27) ? sw : 0);   // rows of the staged rectangle
      const int xa = (int)(tx & 0xffff), xb = xa + (int)(tx >> 27);
      const unsigned top = in[o0 + xa] * (2048 - fx) + in[o0 + xb] * fx;
      const unsigned bot = in[o1 + xa] * (2048 - fx) + in[o1 + xb] * fx;
      const uint8_t v = (uint8_t)((top * (2048 - fy) + bot * fy + (1u << 21)) >> 22);   // < 2^31: exact in 32 bits
      out[i] = v;
      img[(size_t)y * W + x] = v;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ FAST-9/16 score
__constant__ int c_circle[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0}, {3, 1}, {2, 2}, {1, 3},
                                    {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};

// FAST-9/16 score: max over the 16 arcs of 9 contiguous circle pixels of min(I_i - p) (bright) or
// min(p - I_i) (dark); the pixel is a corner at threshold t iff score > t.
// Any arc of 9 contains at least two of the four compass pixels (0, 4, 8, 12), so a pixel with
// fewer than two compass pixels beyond the threshold on one side cannot be a corner: that test
// rejects most of the image after 5 loads. The arc minima are built by doubling (1,2,4,8,+1).
__device__ __forceinline__ bool fast_pretest(const uint8_t* __restrict__ c, int w, int thr) {
  const int p = c[0];
  const int d0 = (int)c[-3 * w] - p, d4 = (int)c[3] - p, d8 = (int)c[3 * w] - p, d12 = (int)c[-3] - p;
  const int nb = (d0 > thr) + (d4 > thr) + (d8 > thr) + (d12 > thr);
  const int nd = (d0 < -thr) + (d4 < -thr) + (d8 < -thr) + (d12 < -thr);
  return nb >= 2 || nd >= 2;
}

// The arc search proper (for pixels that passed fast_pretest).
__device__ __forceinline__ int fast_score_full(const uint8_t* __restrict__ c, int w, int thr) {
  const int p = c[0];
  int d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) d[i] = (int)c[c_circle[i][1] * w + c_circle[i][0]] - p;
  int lo[16], hi[16];   // running min / max over windows of length 1,2,4,8 starting at k
#pragma unroll
  for (int k = 0; k < 16; ++k) { lo[k] = d[k]; hi[k] = d[k]; }
#pragma unroll
  for (int step = 1; step <= 4; step <<= 1) {
    int l2[16], h2[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { l2[k] = min(lo[k], lo[(k + step) & 15]); h2[k] = max(hi[k], hi[(k + step) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) { lo[k] = l2[k]; hi[k] = h2[k]; }
  }
  int best = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int mb = min(lo[k], d[(k + 8) & 15]);       // min over 9 -> bright arc strength
    const int md = -max(hi[k], d[(k + 8) & 15]);      // min over 9 of (p - I) -> dark arc strength
    best = max(best, max(mb, md));
  }
  return best > thr ? min(best, 255) : 0;
}

struct Cand {
  long key;   // Harris response 25(ab - c^2) - (a+b)^2, exact integer
  int pix;    // y * w + x at its level
  int score;  // FAST score (1..255)
};

// FAST score + 3x3 non-maximum suppression + foreground test in one pass: a block scores its
// 64x16 tile plus a 1-pixel ring into LDS (the score image never exists in memory), keeps the
// pixels whose score is strictly greater than all 8 neighbours and whose level-0 position is not
// inside a (grown) vehicle box, appends them to the level's candidate list (order is irrelevant: the selection is by
// value) and counts them in the level's 256-bin FAST-score histogram.
constexpr int kTileW = 64, kTileH = 16;
constexpr int kTileRects = 256;     // rectangles one tile's footprint can meet (a level-7 tile covers ~230 x 57 level-0 pixels)
__global__ __launch_bounds__(256) void fast_detect_kernel(const int4* __restrict__ rects, int n_rects, int w0, int h0, const Levels L, int thr,
                                                          Cand* __restrict__ cand, int* __restrict__ cand_n, int* __restrict__ score_hist) {
  __shared__ uint8_t s_sc[(kTileH + 2) * (kTileW + 2)];
  __shared__ uint8_t s_px[(kTileH + 8) * (kTileW + 8)];   // tile + 4: 1 (NMS ring) + 3 (circle radius)
  int li = 0;
#pragma unroll
  for (int i = 1; i < kPyrLevels; ++i)
    if (i < L.n && (int)blockIdx.x >= L.l[i].tile_begin) li = i;
  const Level lv = L.l[li];
  const int t = blockIdx.x - lv.tile_begin;
  const int tx0 = (t % lv.tiles_x) * kTileW, ty0 = (t / lv.tiles_x) * kTileH;
  constexpr int PP = kTileW + 8;
  {
    // Every load of the tile is issued before the first one is stored: as a plain loop (load, store to LDS, next) the seven byte
    // loads of a thread were seven dependent round trips to memory, ~15 us of a workgroup's life and most of this kernel's 58 us.
    constexpr int NPX = (kTileH + 8) * PP, NIT = (NPX + 255) / 256;
    uint8_t v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = min((int)threadIdx.x + 256 * k, NPX - 1);
      const int x = min(max(tx0 - 4 + i % PP, 0), lv.w - 1), y = min(max(ty0 - 4 + i / PP, 0), lv.h - 1);
      v[k] = lv.img[(size_t)y * lv.w + x];       // clamped addresses; clamped pixels are never used by a scored pixel
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if ((int)threadIdx.x + 256 * k < NPX) s_px[threadIdx.x + 256 * k] = v[k];
  }
  __syncthreads();
  // phase A: the 5-load compass test on every pixel; the survivors (a scattered minority) are listed so that
  // phase B runs the 16-pixel arc search densely instead of dragging whole waves through it
  __shared__ unsigned short s_list[(kTileH + 2) * (kTileW + 2)];
  __shared__ int s_nlist, s_ncand, s_base, s_nrect;
  __shared__ int4 s_rect[kTileRects];
  __shared__ int s_cpix[kTileH * kTileW];
  __shared__ uint8_t s_csc[kTileH * kTileW];
  __shared__ int s_hist[256];
  s_hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_nlist = 0; s_ncand = 0; s_nrect = 0; }
  __syncthreads();
  if (n_rects > 0) {
    // rectangles (level-0 pixels, inclusive) that intersect this tile's level-0 footprint: usually none or a few of ~130;
    // the mask used to be an image the rectangles were drawn into and erased from around this kernel (two more launches)
    const int fx0 = (int)(((long)tx0 * w0 + lv.w / 2) / lv.w), fx1 = min((int)(((long)(tx0 + kTileW - 1) * w0 + lv.w / 2) / lv.w), w0 - 1);
    const int fy0 = (int)(((long)ty0 * h0 + lv.h / 2) / lv.h), fy1 = min((int)(((long)(ty0 + kTileH - 1) * h0 + lv.h / 2) / lv.h), h0 - 1);
    for (int k = threadIdx.x; k < n_rects; k += 256) {
      const int4 r = rects[k];
      if (r.x <= fx1 && r.z >= fx0 && r.y <= fy1 && r.w >= fy0) {
        const int slot = atomicAdd(&s_nrect, 1);
        if (slot < kTileRects) s_rect[slot] = r;
      }
    }
  }
  for (int i = threadIdx.x; i < (kTileH + 2) * (kTileW + 2); i += 256) {
    const int lx = i % (kTileW + 2), ly = i / (kTileW + 2);
    const int x = tx0 - 1 + lx, y = ty0 - 1 + ly;
    s_sc[i] = 0;
    if (x >= kBorder && x < lv.w - kBorder && y >= kBorder && y < lv.h - kBorder && fast_pretest(s_px + (ly + 3) * PP + lx + 3, PP, thr))
      s_list[atomicAdd(&s_nlist, 1)] = (unsigned short)i;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < s_nlist; k += 256) {
    const int i = s_list[k];
    const int lx = i % (kTileW + 2), ly = i / (kTileW + 2);
    s_sc[i] = (uint8_t)fast_score_full(s_px + (ly + 3) * PP + lx + 3, PP, thr);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kTileH * kTileW; i += 256) {
    const int lx = i % kTileW, ly = i / kTileW;
    const uint8_t* c = s_sc + (ly + 1) * (kTileW + 2) + lx + 1;
    const int sv = c[0];
    if (sv == 0) continue;
    constexpr int S = kTileW + 2;
    if (!(sv > c[-S - 1] && sv > c[-S] && sv > c[-S + 1] && sv > c[-1] && sv > c[1] && sv > c[S - 1] && sv > c[S] && sv > c[S + 1])) continue;
    const int x = tx0 + lx, y = ty0 + ly;
    if (s_nrect > 0) {      // the foreground test: the corner's level-0 position against the (grown) vehicle boxes that reach this tile
      const int x0 = min((int)(((long)x * w0 + lv.w / 2) / lv.w), w0 - 1), y0 = min((int)(((long)y * h0 + lv.h / 2) / lv.h), h0 - 1);
      bool masked = false;
      if (s_nrect <= kTileRects) {
        for (int k = 0; k < s_nrect && !masked; ++k) {
          const int4 r = s_rect[k];
          masked = x0 >= r.x && x0 <= r.z && y0 >= r.y && y0 <= r.w;
        }
      } else {                                     // more boxes over this tile than the list holds: test against all of them
        for (int k = 0; k < n_rects && !masked; ++k) {
          const int4 r = rects[k];
          masked = x0 >= r.x && x0 <= r.z && y0 >= r.y && y0 <= r.w;
        }
      }
      if (masked) continue;
    }
    const int k = atomicAdd(&s_ncand, 1);                    // workgroup-local list: one global reservation per tile
    s_cpix[k] = y * lv.w + x;
    s_csc[k] = (uint8_t)sv;
  }
  __syncthreads();
  const int nloc = s_ncand;
  if (nloc == 0) return;
  const int sub = t % kCandSub;
  if (threadIdx.x == 0) s_base = atomicAdd(&cand_n[li * kCandSub + sub], nloc);
  __syncthreads();
  const int base = s_base;
  for (int k = threadIdx.x; k < nloc; k += 256) {
    if (base + k < lv.sub_cap) {
      Cand cd;
      cd.key = 0;
      cd.pix = s_cpix[k];
      cd.score = s_csc[k];
      cand[lv.cand_off + sub * lv.sub_cap + base + k] = cd;
      atomicAdd(&s_hist[s_csc[k]], 1);
    }
  }
  __syncthreads();
  if (s_hist[threadIdx.x] != 0) atomicAdd(&score_hist[li * 256 + threadIdx.x], s_hist[threadIdx.x]);
}

// Stage 1 of the selection, as OpenCV's ORB does it: per level keep the 2*n_want candidates with
// the best FAST score (all candidates tied with the last one included), i.e. those with
// score >= cutoff where cutoff is read off the level's score histogram. Stage 2 input: the Harris
// response of every kept candidate (7x7 block of 3x3 Sobel derivatives, exact integers), appended
// to the eligible list. One thread per candidate: the 9x9 neighbourhood streams through a 3-row
// register window, so a candidate costs 81 byte loads and ~600 integer ops with no cross-lane
// traffic, and thousands of candidates are in flight at once.
__global__ __launch_bounds__(256) void harris_kernel(const Levels L, const Cand* __restrict__ cand, const int* __restrict__ cand_n,
                                                     const int* __restrict__ score_hist, Cand* __restrict__ elig,
                                                     int* __restrict__ elig_n) {
  // cutoff per level: smallest score whose suffix count reaches 2 * n_want. The histogram comes into LDS with
  // one coalesced pass (a serial walk over global memory cost ~50 us of pure latency in every workgroup);
  // wave w < L.n then scans level w: lane l owns bins 4l..4l+3, suffix sums by shuffles.
  __shared__ int s_cut[kPyrLevels];
  __shared__ int s_h[kPyrLevels * 256];
  for (int i = threadIdx.x; i < L.n * 256; i += blockDim.x) s_h[i] = score_hist[i];
  __syncthreads();
  {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int li = wv; li < L.n; li += (int)(blockDim.x >> 6)) {
      const int need = 2 * L.l[li].n_want;
      const int* hh = s_h + li * 256 + 4 * lane;
      const int c0 = lane == 0 ? 0 : hh[0], c1 = hh[1], c2 = hh[2], c3 = hh[3];     // bin 0 (no corner) is not counted
      int suf = c0 + c1 + c2 + c3;                       // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_down(suf, o);
        if (lane + o < 64) suf += v;
      }
      const int above = suf - (c0 + c1 + c2 + c3);       // counts of the bins above this lane's four
      // largest bin b with count(bins >= b) >= need; bins inside the lane from the top
      int cut = 0;
      int acc = above + c3;
      if (acc >= need) cut = 4 * lane + 3;
      else if ((acc += c2) >= need) cut = 4 * lane + 2;
      else if ((acc += c1) >= need) cut = 4 * lane + 1;
      else if ((acc += c0) >= need) cut = 4 * lane;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) cut = max(cut, __shfl_xor(cut, o));
      if (lane == 0) s_cut[li] = max(cut, 1);
    }
  }
  __syncthreads();
  const int nthreads = gridDim.x * blockDim.x;
  for (int li = 0; li < L.n; ++li) {
    const Level lv = L.l[li];
    const int cut = s_cut[li];
    for (int sb = 0; sb < kCandSub; ++sb) {
    const int n = min(cand_n[li * kCandSub + sb], lv.sub_cap);
    const Cand* __restrict__ src = cand + lv.cand_off + sb * lv.sub_cap;
    // which workgroups take a sub-list's first chunks rotates with the sub-list, so the short lists spread over the grid
    const int first = (int)((blockIdx.x + 37u * (unsigned)(li * kCandSub + sb)) % gridDim.x);
    for (int ib = first * (int)blockDim.x; ib < n; ib += nthreads) {     // block-uniform trip count: the append below is wave-wide
      const int i = ib + (int)threadIdx.x;
      Cand cd{};
      bool take = i < n;
      if (take) { cd = src[i]; take = cd.score >= cut; }
      if (take) {
      const int x = cd.pix % lv.w, y = cd.pix / lv.w;
      const uint8_t* base = lv.img + (size_t)(y - 4) * lv.w + x - 4;
      int r0[9], r1[9], r2[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { r0[k] = base[k]; r1[k] = base[lv.w + k]; }
      long a = 0, b = 0, c = 0;
#pragma unroll
      for (int row = 2; row < 9; ++row) {
        const uint8_t* rp = base + (size_t)row * lv.w;
#pragma unroll
        for (int k = 0; k < 9; ++k) r2[k] = rp[k];
        int sa = 0, sb = 0, sc = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
          const int ix = (r0[k + 1] + 2 * r1[k + 1] + r2[k + 1]) - (r0[k - 1] + 2 * r1[k - 1] + r2[k - 1]);
          const int iy = (r2[k - 1] + 2 * r2[k] + r2[k + 1]) - (r0[k - 1] + 2 * r0[k] + r0[k + 1]);
          sa += ix * ix; sb += iy * iy; sc += ix * iy;
        }
        a += sa; b += sb; c += sc;
#pragma unroll
        for (int k = 0; k < 9; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; }
      }
      cd.key = 25 * (a * b - c * c) - (a + b) * (a + b);
      }
      const int slot = gtx_wave_append(&elig_n[li], take);
      if (take && slot < lv.cand_cap) elig[lv.cand_off + slot] = cd;
    }
    }
  }
}

// ------------------------------------------------------------------ top-N per level
// One workgroup per level: 8-pass MSB radix select on the (sign-flipped) 64-bit key finds the
// key of the n-th best candidate; candidates above it are kept, ties on the boundary key are
// broken by the smaller pixel index; the kept set is then ordered by (key desc, pix asc) with a
// counting rank so the output order is deterministic.
struct KeyPoint {
  int x, y;      // level pixel
  int level;
  int bin;       // orientation bin (filled by the describe kernel)
};

__device__ __forceinline__ unsigned long long flip_key(long k) { return (unsigned long long)k ^ 0x8000000000000000ull; }

constexpr int kSortCap = 8192;   // eligible candidates the in-LDS sort handles (more: radix-select path)
constexpr int kTopCap = 2048;    // keypoints per level the radix-select path keeps (level 0 takes ~22 % of max_features)

// Stage 2 of the selection, one workgroup per level: the n_want best Harris keys of the eligible set, ordered by
// (key desc, pix asc). Usual case (n <= 8192): one bitonic sort of {key, pix} pairs in LDS (select_by_sort). More eligible
// candidates than that: 8-pass MSB radix select + counting rank (select_by_radix). One launch for both (round 4: they used to
// be two launches that each returned at once for the levels of the other).
__device__ void select_by_sort(unsigned char* s_raw, const Cand* __restrict__ c, int n, int want, const Level& lv, int li,
                               KeyPoint* __restrict__ kps, int* __restrict__ kp_n) {
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_raw);        // [kSortCap]
  int* s_pix = reinterpret_cast<int*>(s_raw + sizeof(unsigned long long) * kSortCap);  // [kSortCap]
  const int tid = threadIdx.x;
  int P = 64;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += blockDim.x) {
    s_key[i] = i < n ? flip_key(c[i].key) : 0ull;
    s_pix[i] = i < n ? c[i].pix : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < P / 2; t += blockDim.x) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool first_block = ((lo & k) == 0);  // this pair sorts "best first"
        const unsigned long long ka = s_key[lo], kb = s_key[hi];
        const int pa = s_pix[lo], pb = s_pix[hi];
        const bool a_before_b = ka > kb || (ka == kb && pa < pb);
        if (a_before_b != first_block) { s_key[lo] = kb; s_key[hi] = ka; s_pix[lo] = pb; s_pix[hi] = pa; }
      }
      __syncthreads();
    }
  for (int i = tid; i < want; i += blockDim.x) {
    KeyPoint kp;
    kp.x = s_pix[i] % lv.w;
    kp.y = s_pix[i] / lv.w;
    kp.level = li;
    kp.bin = 0;
    kps[lv.kp_off + i] = kp;
  }
  if (tid == 0) kp_n[li] = want;
}

// General case (more than kSortCap eligible candidates on a level): 8-pass MSB radix select on the
// (sign-flipped) 64-bit key finds the key of the n-th best candidate; candidates above it are kept,
// ties on the boundary key are broken by the smaller pixel index; the kept set is then ordered by
// (key desc, pix asc) with a counting rank.
__device__ void select_by_radix(unsigned char* s_raw, const Cand* __restrict__ c, int n, int want, const Level& lv, int li,
                                KeyPoint* __restrict__ kps, int* __restrict__ kp_n) {
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_raw);                              // [kTopCap] selected set
  int* s_pix = reinterpret_cast<int*>(s_raw + sizeof(unsigned long long) * kTopCap);                     // [kTopCap]
  int* hist = s_pix + kTopCap;                                                                           // [256]
  unsigned long long* s_prefix_p = reinterpret_cast<unsigned long long*>(hist + 256);
  int* s_need_p = reinterpret_cast<int*>(s_prefix_p + 1);
  int* s_nsel_p = s_need_p + 1;
#define s_prefix (*s_prefix_p)
#define s_need (*s_need_p)
#define s_nsel (*s_nsel_p)
  const int tid = threadIdx.x, lane = tid & 63;
  // ---- radix select: find the key of rank `want` (1-based, descending). Harris keys share their
  // leading bytes, so in the first passes every lane of a wave hits the same bin: one LDS atomic per
  // wave when the wave agrees on the digit, per-lane atomics otherwise.
  if (tid == 0) { s_prefix = 0ull; s_need = want; }
  __syncthreads();
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const unsigned long long pmask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
      const int i = i0 + tid;
      int digit = -1;
      if (i < n) {
        const unsigned long long k = flip_key(c[i].key);
        if ((k & pmask) == prefix) digit = (int)((k >> shift) & 255);
      }
      const unsigned long long live = __ballot(digit >= 0);
      if (live) {
        const int first = __shfl(digit, __ffsll((long long)live) - 1, 64);
        const unsigned long long same = __ballot(digit == first);
        if (same == live) {
          if (lane == __ffsll((long long)live) - 1) atomicAdd(&hist[first], __popcll(live));
        } else if (digit >= 0) {
          atomicAdd(&hist[digit], 1);
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int need = s_need, b = 255;
      for (; b > 0; --b) {
        if (hist[b] >= need) break;
        need -= hist[b];
      }
      s_need = need;
      s_prefix = prefix | ((unsigned long long)b << shift);
    }
    __syncthreads();
  }
  const unsigned long long kth = s_prefix;   // key of the want-th best
  const int tie_take = s_need;               // how many candidates with key == kth to keep
  // ---- gather: all keys > kth, plus the `tie_take` smallest-pix candidates with key == kth
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    const unsigned long long k = flip_key(c[i].key);
    bool take = k > kth;
    if (k == kth) {
      int r = 0;   // rank among the (rare) ties by pixel index
      for (int j = 0; j < n; ++j)
        if (flip_key(c[j].key) == kth && c[j].pix < c[i].pix) ++r;
      take = r < tie_take;
    }
    if (take) {
      const int sl = atomicAdd(&s_nsel, 1);
      s_key[sl] = k;
      s_pix[sl] = c[i].pix;
    }
  }
  __syncthreads();
  const int m = s_nsel;   // == want
  // ---- deterministic order: rank by (key desc, pix asc)
  for (int i = tid; i < m; i += blockDim.x) {
    const unsigned long long k = s_key[i];
    const int p = s_pix[i];
    int r = 0;
    for (int j = 0; j < m; ++j) {
      const unsigned long long kj = s_key[j];
      r += (kj > k || (kj == k && s_pix[j] < p)) ? 1 : 0;
    }
    KeyPoint kp;
    kp.x = p % lv.w;
    kp.y = p / lv.w;
    kp.level = li;
    kp.bin = 0;
    kps[lv.kp_off + r] = kp;
  }
  if (tid == 0) kp_n[li] = m;
#undef s_prefix
#undef s_need
#undef s_nsel
}

__global__ __launch_bounds__(1024) void select_kernel(const Cand* __restrict__ cand, const int* __restrict__ cand_n,
                                                      const Levels L, KeyPoint* __restrict__ kps, int* __restrict__ kp_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];            // kSortCap * 12 bytes
  const int li = blockIdx.x;
  const Level lv = L.l[li];
  const int n = min(cand_n[li], lv.cand_cap);
  const Cand* c = cand + lv.cand_off;
  const bool by_sort = n <= kSortCap;
  const int want = by_sort ? min(lv.n_want, n) : min(min(lv.n_want, n), kTopCap);
  if (want == 0) {
    if (threadIdx.x == 0) kp_n[li] = 0;
    return;
  }
  if (by_sort) select_by_sort(s_raw, c, n, want, lv, li, kps, kp_n);
  else select_by_radix(s_raw, c, n, want, lv, li, kps, kp_n);
}

// ------------------------------------------------------------------ orientation + descriptor
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ int c_gauss[7] = {18, 34, 49, 54, 49, 34, 18};
// tan((j + 0.5) * 2*pi/256) * 2^24, j = 0..31: boundaries between orientation bins in the first octant
__constant__ long c_tan[32];

__device__ __forceinline__ int angle_bin(int m10, int m01) {
  const long ax = m10 < 0 ? -(long)m10 : m10, ay = m01 < 0 ? -(long)m01 : m01;
  const bool swap = ay > ax;
  const long hi = swap ? ay : ax, lo = swap ? ax : ay;
  int o = 0;
  if (hi > 0) {
#pragma unroll
    for (int j = 0; j < 32; ++j) o += ((lo << 24) >= hi * c_tan[j]) ? 1 : 0;
  }
  if (swap) o = 64 - o;
  if (m10 < 0) o = 128 - o;
  if (m01 < 0) o = -o;
  return o & (kAngleBins - 1);
}

// One wave per keypoint. The 41x41 patch is staged in wave-private LDS, the intensity centroid
// gives the orientation bin, the 35x35 interior is blurred (separable integer 7-tap Gaussian) in
// LDS, and the 256 steered-BRIEF tests are evaluated 64 at a time: a ballot per group of 64 tests
// is one 64-bit word of the descriptor.
// Output goes straight to the dense per-frame lists (level by level, rank order inside a level: a slot's place is the
// keypoint counts of the levels before it + its rank), and the pass's candidate counters / score histograms, which nobody
// reads after the selection, are cleared here for the next frame: the former compact_kernel launch is gone (round 4).
__global__ __launch_bounds__(256) void describe_kernel(const Levels L,
                                                       const KeyPoint* __restrict__ kps, const int* __restrict__ kp_n,
                                                       const int8_t* __restrict__ pattern /*[bins][256][4]*/,
                                                       KeyPoint* __restrict__ okp, unsigned long long* __restrict__ odesc /*[n][4]*/,
                                                       float2* __restrict__ oxy, int* __restrict__ total, float inv_ratio, int total_slots,
                                                       int* __restrict__ counters, int n_counters) {
  __shared__ uint8_t s_patch[4][kPatchW * kPatchW];
  __shared__ unsigned short s_h[4][kPatchW * (kPatchW - 6)];   // horizontal pass, columns 3..37
  __shared__ uint8_t s_blur[4][(kPatchW - 6) * (kPatchW - 6)];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slot = blockIdx.x * 4 + wv;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_counters; i += gridDim.x * blockDim.x) counters[i] = 0;
  if (slot >= total_slots) return;
  // which level does this slot belong to, and how many keypoints do the levels before it hold?
  int li = 0, base = 0, all = 0;
#pragma unroll
  for (int i = 0; i < kPyrLevels; ++i) {
    const int cnt = i < L.n ? kp_n[i] : 0;
    if (i >= 1 && i < L.n && slot >= L.l[i].kp_off) { li = i; base = all; }
    all += cnt;
  }
  if (slot == 0 && lane == 0) *total = all;
  const Level lv = L.l[li];
  if (slot - lv.kp_off >= kp_n[li]) return;
  const int dst = base + slot - lv.kp_off;
  KeyPoint kp = kps[slot];
  const uint8_t* img = lv.img;
  uint8_t* P = s_patch[wv];
  for (int i = lane; i < kPatchW * kPatchW; i += 64) {
    const int u = i % kPatchW - kPatchR, v = i / kPatchW - kPatchR;
    P[i] = img[(size_t)(kp.y + v) * lv.w + kp.x + u];
  }
  // wave-private LDS: no workgroup barrier is needed, LDS instructions of one wave execute in
  // order; the fence only keeps the compiler from moving reads above the writes.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  int m10 = 0, m01 = 0;
  for (int i = lane; i < 31 * 31; i += 64) {
    const int u = i % 31 - 15, v = i / 31 - 15;
    if (abs(u) <= c_umax[abs(v)]) {
      const int val = P[(v + kPatchR) * kPatchW + u + kPatchR];
      m10 += u * val;
      m01 += v * val;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    m10 += __shfl_xor(m10, o, 64);
    m01 += __shfl_xor(m01, o, 64);
  }
  const int bin = angle_bin(m10, m01);
  // blur: horizontal pass over all 41 rows, columns -17..17
  constexpr int BW = kPatchW - 6;   // 35
  unsigned short* Hh = s_h[wv];
  for (int i = lane; i < kPatchW * BW; i += 64) {
    const int r = i / BW, cidx = i % BW;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += c_gauss[k] * P[r * kPatchW + cidx + k];
    Hh[i] = (unsigned short)acc;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint8_t* Bl = s_blur[wv];
  for (int i = lane; i < BW * BW; i += 64) {
    const int r = i / BW, cidx = i % BW;
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += c_gauss[k] * Hh[(r + k) * BW + cidx];
    Bl[i] = (uint8_t)((acc + 32768) >> 16);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int8_t* pat = pattern + (size_t)bin * 256 * 4;
  unsigned long long* d = odesc + (size_t)dst * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int t = g * 64 + lane;
    const char4 q = reinterpret_cast<const char4*>(pat)[t];
    const int va = Bl[(q.y + 17) * BW + q.x + 17], vb = Bl[(q.w + 17) * BW + q.z + 17];
    const unsigned long long word = __ballot(va < vb);
    if (lane == 0) d[g] = word;
  }
  if (lane == 0) {
    kp.bin = bin;
    okp[dst] = kp;
    oxy[dst] = make_float2((float)kp.x * lv.scale * inv_ratio, (float)kp.y * lv.scale * inv_ratio);
  }
}

// ------------------------------------------------------------------ matching
// Brute-force Hamming 2-NN + Lowe ratio test + ordered compaction in ONE launch (round 4; two launches before).
// Grid = (query tiles of 256) x (train chunks of 256): a block stages its train chunk in LDS and every thread scans it for
// its query, leaving a partial (best, second, index) per chunk. The workgroup that finishes LAST (a ticket drawn behind a
// workgroup barrier, after every wave has waited for its stores) merges the chunks of every query in chunk order -- which
// preserves the "lowest index wins ties" rule --, applies the ratio test and compacts the survivors in query order.
// Cross-workgroup visibility (MI355X_MICROARCH.md, hand-off table, first row): the partials are written with write-through
// (`sc1`) stores and read back with `sc1` loads, each storing wave drains its stores (`s_waitcnt vmcnt(0)`) before the
// barrier, ONE lane per workgroup then adds to the device-scope ticket, and only the workgroup whose add returned the
// last ticket reads, behind its own barrier. No cache-wide write-back or invalidate is involved.
constexpr int kMatchChunk = 256;
__device__ __forceinline__ void st_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_sc1(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void match_kernel(const unsigned long long* __restrict__ q, const int* __restrict__ nq_p,
                                                    const unsigned long long* __restrict__ t, const int* __restrict__ nt_p,
                                                    int max_q, int* __restrict__ part_idx, int* __restrict__ part_d1,
                                                    int* __restrict__ part_d2, unsigned* __restrict__ ticket,
                                                    float ratio, int keep_all, const float2* __restrict__ q_xy, const float2* __restrict__ t_xy,
                                                    int* __restrict__ best_idx, int* __restrict__ best_d, int* __restrict__ second_d,
                                                    int* __restrict__ m_q, int* __restrict__ m_t, int* __restrict__ m_d,
                                                    float4* __restrict__ m_pts, int* __restrict__ n_match) {
  __shared__ unsigned long long s_t[kMatchChunk * 4];
  __shared__ int s_last;
  const int nq = *nq_p, nt = *nt_p;
  const int t0 = blockIdx.y * kMatchChunk;
  const int tid = threadIdx.x;
  const bool active = blockIdx.x * blockDim.x < nq && t0 < nt;     // whole workgroup: uniform
  if (active) {
    const int i = blockIdx.x * blockDim.x + tid;
    const int j = t0 + tid;
    if (j < nt) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s_t[tid * 4 + k] = t[(size_t)j * 4 + k];
    }
    __syncthreads();
    if (i < nq) {
      const unsigned long long a0 = q[(size_t)i * 4], a1 = q[(size_t)i * 4 + 1], a2 = q[(size_t)i * 4 + 2], a3 = q[(size_t)i * 4 + 3];
      int b1 = 1 << 30, b2 = 1 << 30, bi = -1;
      const int lim = min(kMatchChunk, nt - t0);
      for (int k = 0; k < lim; ++k) {
        const int d = __popcll(a0 ^ s_t[k * 4]) + __popcll(a1 ^ s_t[k * 4 + 1]) + __popcll(a2 ^ s_t[k * 4 + 2]) +
                      __popcll(a3 ^ s_t[k * 4 + 3]);
        if (d < b1) { b2 = b1; b1 = d; bi = t0 + k; }
        else if (d < b2) b2 = d;
      }
      const size_t o = (size_t)blockIdx.y * max_q + i;
      st_sc1(&part_idx[o], bi); st_sc1(&part_d1[o], b1); st_sc1(&part_d2[o], b2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's partials have left the CU
  __syncthreads();
  if (tid == 0) {
    const unsigned drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = drawn == gridDim.x * gridDim.y - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: merge of the per-chunk partials + Lowe ratio test + ordered compaction (queries in index order)
  int* s_cnt = reinterpret_cast<int*>(s_t);                        // [256] (+ 4 wave totals behind them)
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next pass
  const int nchunk = (nt + kMatchChunk - 1) / kMatchChunk;
  for (int i = tid; i < nq; i += blockDim.x) {
    int b1 = 1 << 30, b2 = 1 << 30, bi = -1;
    // eight chunks' partials are requested before the first is looked at: one round trip per eight chunks instead of
    // two dependent ones per chunk (this single workgroup's merge was 60 of the launch's 84 us)
    for (int c0 = 0; c0 < nchunk; c0 += 8) {
      int d1[8], d2[8], ix[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t o = (size_t)min(c0 + k, nchunk - 1) * max_q + i;
        d1[k] = ld_sc1(&part_d1[o]); d2[k] = ld_sc1(&part_d2[o]); ix[k] = ld_sc1(&part_idx[o]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (c0 + k >= nchunk) break;
        if (d1[k] < b1) { b2 = min(b1, d2[k]); b1 = d1[k]; bi = ix[k]; }
        else b2 = min(b2, d1[k]);
      }
    }
    best_idx[i] = bi; best_d[i] = b1; second_d[i] = b2;
  }
  __syncthreads();
  const int per = (nq + 255) / 256;
  const int lo = tid * per, hi = min(lo + per, nq);
  int c = 0;
  for (int i = lo; i < hi; ++i)
    if (keep_all ? (nt >= 1 && best_idx[i] >= 0) : (nt >= 2 && (float)best_d[i] < ratio * (float)second_d[i])) ++c;
  // exclusive prefix of the 256 per-thread counts: inclusive scan inside each wave by shuffles, then the 4 wave totals
  {
    const int lane = tid & 63, wv = tid >> 6;
    int inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(inc, d, 64);
      if (lane >= d) inc += v;
    }
    if (lane == 63) s_cnt[256 + wv] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = s_cnt[256 + k];
      if (k < wv) base += v;
      total += v;
    }
    s_cnt[tid] = base + inc - c;
    if (tid == 0) *n_match = total;
  }
  __syncthreads();
  int o = s_cnt[tid];
  for (int i = lo; i < hi; ++i)
    if (keep_all ? (nt >= 1 && best_idx[i] >= 0) : (nt >= 2 && (float)best_d[i] < ratio * (float)second_d[i])) {
      m_q[o] = i; m_t[o] = best_idx[i]; m_d[o] = best_d[i];
      const float2 a = q_xy[i], b = t_xy[best_idx[i]];
      m_pts[o] = make_float4(a.x, a.y, b.x, b.y);
      ++o;
    }
}

// ------------------------------------------------------------------ RANSAC
__device__ __forceinline__ unsigned hash_u32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// Homography from 4 correspondences by the projective-basis construction: A maps the canonical
// basis (e1, e2, e3, e1+e2+e3) to the four source points, B to the four destination points,
// H = B * adj(A). Closed form, static indexing only (stays in registers), f64.
__device__ __forceinline__ bool basis_matrix(const double* px, const double* py, double* M) {
  // solve [p0 p1 p2] (l0,l1,l2)^T = p3 by Cramer's rule; columns then scaled by l
  const double x0 = px[0], y0 = py[0], x1 = px[1], y1 = py[1], x2 = px[2], y2 = py[2], x3 = px[3], y3 = py[3];
  const double det = x0 * (y1 - y2) - x1 * (y0 - y2) + x2 * (y0 - y1);
  if (!(fabs(det) > 1e-9)) return false;
  const double l0 = (x3 * (y1 - y2) - x1 * (y3 - y2) + x2 * (y3 - y1)) / det;
  const double l1 = (x0 * (y3 - y2) - x3 * (y0 - y2) + x2 * (y0 - y3)) / det;
  const double l2 = (x0 * (y1 - y3) - x1 * (y0 - y3) + x3 * (y0 - y1)) / det;
  if (!(fabs(l0) > 1e-9 && fabs(l1) > 1e-9 && fabs(l2) > 1e-9)) return false;   // three points collinear
  M[0] = l0 * x0; M[1] = l1 * x1; M[2] = l2 * x2;
  M[3] = l0 * y0; M[4] = l1 * y1; M[5] = l2 * y2;
  M[6] = l0;      M[7] = l1;      M[8] = l2;
  return true;
}
__device__ bool homography4(const double* px, const double* py, const double* qx, const double* qy, double* H) {
  double A[9], B[9];
  if (!basis_matrix(px, py, A) || !basis_matrix(qx, qy, B)) return false;
  const double adj[9] = {A[4] * A[8] - A[5] * A[7], A[2] * A[7] - A[1] * A[8], A[1] * A[5] - A[2] * A[4],
                         A[5] * A[6] - A[3] * A[8], A[0] * A[8] - A[2] * A[6], A[2] * A[3] - A[0] * A[5],
                         A[3] * A[7] - A[4] * A[6], A[1] * A[6] - A[0] * A[7], A[0] * A[4] - A[1] * A[3]};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) H[r * 3 + c] = B[r * 3] * adj[c] + B[r * 3 + 1] * adj[3 + c] + B[r * 3 + 2] * adj[6 + c];
  if (!(fabs(H[8]) > 1e-12)) return false;
  const double inv = 1.0 / H[8];
#pragma unroll
  for (int i = 0; i < 9; ++i) H[i] *= inv;
  return true;
}

// Affine map from 3 correspondences (Cramer's rule), as a 3x3 matrix with last row (0, 0, 1).
__device__ bool affine3(const double* px, const double* py, const double* qx, const double* qy, double* H) {
  const double x0 = px[0], y0 = py[0], x1 = px[1], y1 = py[1], x2 = px[2], y2 = py[2];
  const double det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1);
  if (!(fabs(det) > 1e-9)) return false;           // the three source points are collinear
  const double id = 1.0 / det;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const double* q = r == 0 ? qx : qy;
    const double u0 = q[0], u1 = q[1], u2 = q[2];
    H[r * 3 + 0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) * id;
    H[r * 3 + 1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) * id;
    H[r * 3 + 2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) * id;
  }
  H[6] = 0.0; H[7] = 0.0; H[8] = 1.0;
  return true;
}

// One hypothesis: 4 (projective) or 3 (affine) distinct matches drawn by a counter-based hash of (seed, hypothesis, draw),
// exact minimal solve in normalised coordinates, de-normalised H. Straight-line f64 code without LDS: every lane of a wave
// runs it on the same inputs and gets the same bits, which is how the scoring wave below obtains its hypothesis.
__device__ __forceinline__ bool make_hypothesis(const float4* __restrict__ pts, int n, unsigned seed, int hyp, double cx, double cy, double sc, int affine, double H[9]) {
  // inlined, and every small array indexed by unrolled constants: as a call with H behind a pointer and idx[] indexed by a
  // run-time k the kernel kept 80 bytes of scratch per lane -- 9.2 MB of HBM writes per launch for a one-record result
  // (profiles/r05_pmc_traffic.json)
  const int ns = affine ? 3 : 4;
  if (n < ns) return false;
  int idx[4] = {0, 0, 0, 0};
  unsigned ctr = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= ns) break;
    for (;;) {
      const int cand = (int)(hash_u32(seed ^ hash_u32((unsigned)hyp * 977u + ctr)) % (unsigned)n);
      ++ctr;
      bool dup = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) dup |= (j < k) && idx[j] == cand;
      if (!dup) { idx[k] = cand; break; }
    }
  }
  double px[4], py[4], qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 p = pts[idx[k]];
    px[k] = ((double)p.x - cx) * sc; py[k] = ((double)p.y - cy) * sc;
    qx[k] = ((double)p.z - cx) * sc; qy[k] = ((double)p.w - cy) * sc;
  }
  double Hn[9];
  if (!(affine ? affine3(px, py, qx, qy, Hn) : homography4(px, py, qx, qy, Hn))) return false;
  // de-normalise: H = T^-1 Hn T with T = [[sc,0,-sc*cx],[0,sc,-sc*cy],[0,0,1]]
  const double is = 1.0 / sc;
  double M[9];
  for (int r = 0; r < 3; ++r) {   // M = Hn T
    M[r * 3 + 0] = Hn[r * 3 + 0] * sc;
    M[r * 3 + 1] = Hn[r * 3 + 1] * sc;
    M[r * 3 + 2] = Hn[r * 3 + 2] - sc * (Hn[r * 3 + 0] * cx + Hn[r * 3 + 1] * cy);
  }
  for (int k = 0; k < 3; ++k) {   // H = T^-1 M, T^-1 = [[is,0,cx],[0,is,cy],[0,0,1]]
    H[0 + k] = is * M[0 + k] + cx * M[6 + k];
    H[3 + k] = is * M[3 + k] + cy * M[6 + k];
    H[6 + k] = M[6 + k];
  }
  return fabs(H[8]) > 1e-12;
}

// Device-side result record of one stabilize pass, fetched with a single D2H copy (the match points follow it in the same
// buffer: StabOut).
struct StabResult {
  int n_match, best, n_cur, pad;
  double H[9];
};

// Hypothesise, score and pick the winner in ONE launch (round 4; three launches before: solve, score, argmin).
// A wave makes hypothesis `hyp` (every lane the same solve) and scores it: truncated squared reprojection error (MSAC) over
// all matches, quantised to 1/1024 px^2 so that the sum is an exact integer regardless of the reduction order. The
// workgroup's 8 costs are reduced in LDS to one key (cost << 16 | hyp: the lowest cost, then the lowest hypothesis index,
// exactly the order of the former argmin) and folded into `state[0]` with one device-scope atomic minimum; `state[1]` is a
// ticket: the workgroup that draws the last one knows every minimum has been applied (each workgroup's ticket add depends on
// the value its minimum returned), reads the winner back with another atomic, re-derives its H from the index, writes the
// result record and re-arms state for the next pass. Device-scope atomics are performed beyond the per-XCD L2s, so no
// fence and no other cross-workgroup visibility is involved.
constexpr unsigned long long kNoHyp = ~0ull;
__global__ __launch_bounds__(512) void ransac_kernel(const float4* __restrict__ pts, const int* __restrict__ n_p, const int* __restrict__ n_cur_p,
                                                      unsigned seed, int n_hyp, double cx, double cy, double sc, int affine, float thr2,
                                                      unsigned long long* __restrict__ state /*[2]: best key (armed: all ones), tickets (0)*/,
                                                      StabResult* __restrict__ res) {
  constexpr int kHypPerWg = 8;
  __shared__ unsigned long long s_key[kHypPerWg];
  __shared__ int s_last;
  const int n = *n_p;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hyp = blockIdx.x * kHypPerWg + wv;
  unsigned long long key = kNoHyp;
  if (hyp < n_hyp) {
    double H[9];
    if (make_hypothesis(pts, n, seed, hyp, cx, cy, sc, affine, H)) {
      long acc = 0;
      for (int i = lane; i < n; i += 64) {
        const float4 p = pts[i];
        const double x = p.x, y = p.y;
        const double w = H[6] * x + H[7] * y + H[8];
        double e = (double)thr2;
        if (fabs(w) > 1e-12) {
          const double iw = 1.0 / w;
          const double dx = (H[0] * x + H[1] * y + H[2]) * iw - p.z;
          const double dy = (H[3] * x + H[4] * y + H[5]) * iw - p.w;
          e = fmin(dx * dx + dy * dy, (double)thr2);
        }
        acc += (long)(e * 1024.0 + 0.5);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
      key = ((unsigned long long)acc << 16) | (unsigned long long)hyp;     // cost < 2^47 (4096 per match at most), hyp < 2^16
    }
  }
  if (lane == 0) s_key[wv] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long k = s_key[0];
#pragma unroll
    for (int i = 1; i < kHypPerWg; ++i) k = min(k, s_key[i]);
    // The ordering is the memory model's: the minimum is published by the RELEASE half of the ticket's acq_rel read-modify-write
    // (agent scope), and the workgroup that draws the last ticket ACQUIRES every earlier workgroup's release through the
    // ticket word's modification order (a release sequence of RMWs), so its read of state[0] below sees every minimum.
    // The s_waitcnt between the two is not part of that argument: cdna_hip_programming.md (section 6, Guideline 16, Pitfall 12)
    // records that ROCm 7.2 can drop the wait a release implies, and prescribes keeping an explicit one in front of the ticket.
    (void)__hip_atomic_fetch_min(&state[0], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ticket = __hip_atomic_fetch_add(&state[1], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == (unsigned long long)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last workgroup: every minimum has been applied
  if (wv != 0) return;
  unsigned long long best_key = 0;
  if (lane == 0) best_key = __hip_atomic_fetch_min(&state[0], kNoHyp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);   // acquire read-back, in L2
  best_key = __shfl(best_key, 0, 64);
  const int b = best_key == kNoHyp ? -1 : (int)(best_key & 0xffffull);
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (b >= 0) make_hypothesis(pts, n, seed, b, cx, cy, sc, affine, H);
  if (lane == 0) {
    res->n_match = n;
    res->n_cur = *n_cur_p;
    res->best = b;
    for (int k = 0; k < 9; ++k) res->H[k] = H[k];
    atomicExch(&state[0], kNoHyp);                                         // re-arm for the next pass on this stream
    atomicExch(&state[1], 0ull);
  }
}

}  // namespace

// =========================================================================== host side

namespace {

// Steered-BRIEF sampling pattern: 256 point pairs drawn once from an isotropic Gaussian
// (sigma = patch/5, BRIEF "G II"), clipped to |x|,|y| <= 12, fixed seed. OpenCV's learned ORB
// table is not available in the reference tree; descriptor bits are internal to the stabilizer.
void base_pattern(int8_t out[256][4]) {
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {   // splitmix64
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  auto gauss = [&]() {   // Irwin-Hall(12) - 6: integer-friendly approximate normal, exactly reproducible
    long acc = 0;
    for (int i = 0; i < 12; ++i) acc += (long)(next() >> 40);            // 24-bit uniforms
    return ((double)acc / 16777216.0 - 6.0);
  };
  for (int i = 0; i < 256; ++i)
    for (int k = 0; k < 4; ++k) {
      for (;;) {
        const double v = gauss() * 6.2;
        const long r = std::lround(v);
        if (r >= -12 && r <= 12) { out[i][k] = (int8_t)r; break; }
      }
    }
  for (int i = 0; i < 256; ++i)   // a pair must not compare a pixel with itself
    if (out[i][0] == out[i][2] && out[i][1] == out[i][3]) out[i][2] = (int8_t)(out[i][2] >= 12 ? out[i][2] - 1 : out[i][2] + 1);
}

}  // namespace

void stabilizer_pattern_table(std::vector<int8_t>& out) {
  int8_t base[256][4];
  base_pattern(base);
  out.resize((size_t)kAngleBins * 256 * 4);
  for (int b = 0; b < kAngleBins; ++b) {
    const double th = b * (2.0 * M_PI / kAngleBins), cs = std::cos(th), sn = std::sin(th);
    for (int i = 0; i < 256; ++i)
      for (int k = 0; k < 2; ++k) {
        const double x = base[i][2 * k], y = base[i][2 * k + 1];
        out[((size_t)b * 256 + i) * 4 + 2 * k] = (int8_t)std::lround(x * cs - y * sn);
        out[((size_t)b * 256 + i) * 4 + 2 * k + 1] = (int8_t)std::lround(x * sn + y * cs);
      }
  }
}

struct Stabilizer::Impl {
  gtx_ctx* ctx;
  gtx_stab_config cfg;
  int fh, fw;            // frame
  int gh, gw;            // level-0 gray (frame * downsample_ratio)
  int half;              // 1 when downsample_ratio == 0.5
  Levels lev_ref{}, lev_cur{};
  int pyr_bytes = 0, cand_total = 0, slots_ref = 0, slots_cur = 0;
  int n_hyp = 0;
  std::vector<int8_t> pattern;   // [bins][256][4]

  DevBuf d_frame, d_pyr, d_pyr_tab, d_clahe_lut, d_rects, d_cand, d_elig, d_counters, d_kp_n, d_kps, d_pattern;
  struct Feat {
    DevBuf kps, desc, xy, n;
    int host_n = 0;
  } ref, cur;
  DevBuf d_bidx, d_bd, d_sd, d_mq, d_mt, d_md, d_nmatch, d_best, d_pidx, d_pd1, d_pd2, d_rstate, d_mticket;
  DevBuf d_out;                  // StabResult, then the match points (float4 x slots): one D2H copy per pass
  uint8_t* h_out = nullptr;      // pinned image of d_out
  StabResult* h_res = nullptr;   // = h_out
  float4* h_pts = nullptr;       // = h_out + kOutPts
  static constexpr size_t kOutPts = (sizeof(StabResult) + 15) / 16 * 16;
  StabResult* d_res() const { return d_out.as<StabResult>(); }
  float4* d_mpts() const { return reinterpret_cast<float4*>(d_out.as<uint8_t>() + kOutPts); }
  hipEvent_t done_ev = nullptr;
  hipEvent_t t0_ev = nullptr, t1_ev = nullptr;   // GPU time of the last submitted pass (timing events)
  float last_ms = 0.f;
  bool timed = false;                            // t0_ev was recorded for the pass in flight
  bool pending = false;
  bool have_ref = false;
  // last results
  double H[9];
  bool valid = false;
  int stats[4] = {0, 0, 0, 0};

  void plan(Levels& L, int max_features, int& slots);
  void build_rects(const float* boxes, int n, std::vector<int4>& rects) const;
  void extract(const uint8_t* gray_dev, const float* boxes, int n, const Levels& L, int slots, Feat& out);
  void gray_from_frame(const uint8_t* frame, int h, int w);
  void submit_match();
  void collect(double Hout[9], int* valid_out, int st[4]);
};

void Stabilizer::Impl::plan(Levels& L, int max_features, int& slots) {
  L.n = cfg.n_levels;
  const double factor = 1.0 / cfg.scale_factor;
  double want = max_features * (1.0 - factor) / (1.0 - std::pow(factor, L.n));
  int sum = 0, off = 0, coff = 0, koff = 0;
  for (int i = 0; i < L.n; ++i) {
    Level& lv = L.l[i];
    const double sc = std::pow((double)cfg.scale_factor, i);
    lv.w = (int)std::lround(gw / sc);
    lv.h = (int)std::lround(gh / sc);
    lv.scale = (float)sc;
    lv.off = off;
    off += lv.w * lv.h;
    lv.cand_off = coff;
    lv.sub_cap = cdiv(std::max(4096, lv.w * lv.h / 16), kCandSub);
    lv.cand_cap = lv.sub_cap * kCandSub;
    coff += lv.cand_cap;
    if (i < L.n - 1) {
      lv.n_want = (int)std::lround(want);
      sum += lv.n_want;
      want *= factor;
    } else {
      lv.n_want = std::max(max_features - sum, 0);
    }
    lv.kp_off = koff;
    koff += lv.n_want;
    lv.tiles_x = cdiv(lv.w, 64);
    lv.tile_begin = i == 0 ? 0 : L.l[i - 1].tile_begin + L.l[i - 1].tiles_x * cdiv(L.l[i - 1].h, 16);
  }
  L.n_tiles = L.l[L.n - 1].tile_begin + L.l[L.n - 1].tiles_x * cdiv(L.l[L.n - 1].h, 16);
  for (int i = 0, t = 0; i < L.n; ++i) {   // resize tables of levels 1.. (same for both plans: sizes only)
    L.tab_off[i] = t;
    if (i > 0) t += L.l[i].w + L.l[i].h;
  }
  // Launch groups of the pyramid: 3 levels, then up to 4 at a time (8 levels: 1-3 and 4-7). GTX_PYR_GROUP=n forces groups of n
  // (1 = a launch per level, as before round 4). A group whose largest rectangle would not fit 24 KB of LDS is split.
  {
    int force = 0;
    if (const char* e = getenv("GTX_PYR_GROUP")) force = std::max(1, std::min(atoi(e), kPyrGroupMax));
    L.n_groups = 0;
    int first = 1;
    while (first < L.n) {
      int n = std::min(L.n - first, force ? force : (first == 1 && L.n - 1 > kPyrGroupMax ? 3 : kPyrGroupMax));
      int need = 0;
      for (;; --n) {
        // largest rectangle over the tiles of the group's last level, by the same backward walk as the kernel
        need = 0;
        const Level& ll = L.l[first + n - 1];
        for (int ty = 0; ty < cdiv(ll.h, 16); ++ty)
          for (int tx = 0; tx < cdiv(ll.w, 64); ++tx) {
            int x0 = tx * 64, y0 = ty * 16, x1 = std::min(x0 + 63, ll.w - 1), y1 = std::min(y0 + 15, ll.h - 1);
            for (int k = first + n - 1; k >= first; --k) {
              need = std::max(need, (x1 - x0 + 1) * (y1 - y0 + 1));
              const Level &cur = L.l[k], &prev = L.l[k - 1];
              const bool right = x1 == cur.w - 1, bottom = y1 == cur.h - 1;
              const unsigned a = pyr_src(x0, prev.w, cur.w), b = pyr_src(x1, prev.w, cur.w), c = pyr_src(y0, prev.h, cur.h), d = pyr_src(y1, prev.h, cur.h);
              x0 = (int)(a & 0xffff); y0 = (int)(c & 0xffff);
              x1 = right ? prev.w - 1 : (int)((b & 0xffff) + (b >> 27));
              y1 = bottom ? prev.h - 1 : (int)((d & 0xffff) + (d >> 27));
            }
            need = std::max(need, (x1 - x0 + 1) * (y1 - y0 + 1));
          }
        if (need <= 24 * 1024 || n == 1) break;
      }
      GTX_CHECK(need <= 64 * 1024, "stabilizer: scale_factor %g needs a %d-byte resize rectangle", cfg.scale_factor, need);
      L.grp_first[L.n_groups] = first;
      L.grp_n[L.n_groups] = n;
      L.grp_buf[L.n_groups] = (need + 15) / 16 * 16;
      ++L.n_groups;
      first += n;
    }
  }
  GTX_CHECK(max_features * 0.25 < kTopCap, "stabilizer: at most ~8000 features per image are supported");
  pyr_bytes = off;
  cand_total = coff;
  slots = koff;
}

Stabilizer::Stabilizer(gtx_ctx* ctx, const gtx_stab_config& cfg) : impl_(new Impl) {
  Impl& S = *impl_;
  S.ctx = ctx;
  S.cfg = cfg;
  GTX_CHECK(cfg.frame_h > 0 && cfg.frame_w > 0, "stabilizer: frame size must be given");
  GTX_CHECK(cfg.downsample_ratio == 0.5f || cfg.downsample_ratio == 1.0f, "stabilizer: downsample_ratio must be 0.5 or 1.0 (got %g)", cfg.downsample_ratio);
  GTX_CHECK(cfg.n_levels >= 1 && cfg.n_levels <= kPyrLevels, "stabilizer: n_levels must be in [1,%d]", kPyrLevels);
  GTX_CHECK(cfg.scale_factor > 1.0f, "stabilizer: scale_factor must be > 1");
  GTX_CHECK(cfg.max_features >= 8, "stabilizer: max_features too small");
  S.fh = cfg.frame_h;
  S.fw = cfg.frame_w;
  S.half = cfg.downsample_ratio == 0.5f ? 1 : 0;
  S.gh = S.half ? S.fh / 2 : S.fh;
  S.gw = S.half ? S.fw / 2 : S.fw;
  GTX_HIP(hipSetDevice(ctx->device));
  const int ref_features = (int)std::lround(cfg.max_features * (double)cfg.ref_multiplier);
  S.plan(S.lev_cur, cfg.max_features, S.slots_cur);
  S.plan(S.lev_ref, ref_features, S.slots_ref);
  // Levels smaller than the keypoint border simply yield no keypoints (as in OpenCV's ORB).
  const Level& l0 = S.lev_ref.l[0];
  GTX_CHECK(l0.w > 2 * kBorder + 8 && l0.h > 2 * kBorder + 8, "stabilizer: %dx%d working image is too small", l0.w, l0.h);
  const int slots = std::max(S.slots_ref, S.slots_cur);
  S.d_pyr.alloc(S.pyr_bytes);
  S.d_clahe_lut.alloc(kClaheLutBytes);
  S.d_rects.alloc(sizeof(int4) * kMaxRects);
  S.d_cand.alloc(sizeof(Cand) * (size_t)S.cand_total);
  S.d_elig.alloc(sizeof(Cand) * (size_t)S.cand_total);
  S.d_counters.alloc(sizeof(int) * (kCounterInts));
  GTX_HIP(hipMemset(S.d_counters.p, 0, sizeof(int) * (kCounterInts)));
  S.d_kp_n.alloc(sizeof(int) * kPyrLevels);
  S.d_kps.alloc(sizeof(KeyPoint) * slots);
  for (Impl::Feat* f : {&S.ref, &S.cur}) {
    f->kps.alloc(sizeof(KeyPoint) * slots);
    f->desc.alloc(32 * (size_t)slots);
    f->xy.alloc(sizeof(float2) * slots);
    f->n.alloc(sizeof(int));
    GTX_HIP(hipMemset(f->n.p, 0, sizeof(int)));
  }
  S.d_bidx.alloc(4 * slots); S.d_bd.alloc(4 * slots); S.d_sd.alloc(4 * slots);
  S.d_mq.alloc(4 * slots); S.d_mt.alloc(4 * slots); S.d_md.alloc(4 * slots);
  S.d_nmatch.alloc(sizeof(int));
  S.n_hyp = std::max(64, std::min(cfg.ransac_max_iter, 2048));
  S.d_best.alloc(sizeof(int));
  S.d_out.alloc(Impl::kOutPts + sizeof(float4) * slots);
  GTX_HIP(hipHostMalloc((void**)&S.h_out, Impl::kOutPts + sizeof(float4) * slots));
  S.h_res = reinterpret_cast<StabResult*>(S.h_out);
  S.h_pts = reinterpret_cast<float4*>(S.h_out + Impl::kOutPts);
  S.d_mticket.alloc(sizeof(unsigned));
  GTX_HIP(hipMemset(S.d_mticket.p, 0, sizeof(unsigned)));
  S.d_rstate.alloc(2 * sizeof(unsigned long long));
  {
    const unsigned long long armed[2] = {kNoHyp, 0ull};
    GTX_HIP(hipMemcpy(S.d_rstate.p, armed, sizeof armed, hipMemcpyHostToDevice));
  }
  GTX_HIP(hipEventCreateWithFlags(&S.done_ev, wait_event_flags(false)));
  GTX_HIP(hipEventCreate(&S.t0_ev));
  GTX_HIP(hipEventCreate(&S.t1_ev));
  {
    const size_t parts = (size_t)cdiv(S.slots_ref, kMatchChunk) * S.slots_cur;
    S.d_pidx.alloc(4 * parts); S.d_pd1.alloc(4 * parts); S.d_pd2.alloc(4 * parts);
  }
  stabilizer_pattern_table(S.pattern);   // rotated sampling patterns
  S.d_pattern.alloc(S.pattern.size());
  GTX_HIP(hipMemcpy(S.d_pattern.p, S.pattern.data(), S.pattern.size(), hipMemcpyHostToDevice));
  GTX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kSortCap * 12));
  {
    const Levels& L = S.lev_cur;
    std::vector<unsigned> tab;
    for (int i = 1; i < L.n; ++i) {
      for (int x = 0; x < L.l[i].w; ++x) tab.push_back(pyr_src(x, L.l[i - 1].w, L.l[i].w));
      for (int y = 0; y < L.l[i].h; ++y) tab.push_back(pyr_src(y, L.l[i - 1].h, L.l[i].h));
    }
    S.d_pyr_tab.alloc(sizeof(unsigned) * std::max<size_t>(tab.size(), 1));
    if (!tab.empty()) GTX_HIP(hipMemcpy(S.d_pyr_tab.p, tab.data(), sizeof(unsigned) * tab.size(), hipMemcpyHostToDevice));
  }
  long tans[32];
  for (int j = 0; j < 32; ++j) tans[j] = std::lround(std::tan((j + 0.5) * 2.0 * M_PI / kAngleBins) * 16777216.0);
  GTX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tan), tans, sizeof tans));
}

Stabilizer::~Stabilizer() {
  if (impl_) {
    if (impl_->h_out) (void)hipHostFree(impl_->h_out);
    if (impl_->done_ev) (void)hipEventDestroy(impl_->done_ev);
    if (impl_->t0_ev) (void)hipEventDestroy(impl_->t0_ev);
    if (impl_->t1_ev) (void)hipEventDestroy(impl_->t1_ev);
  }
}

void Stabilizer::Impl::build_rects(const float* boxes, int n, std::vector<int4>& rects) const {
  // stabilo masks each box grown by mask_margin_ratio of its size, drawn on the downsampled frame
  const float r = cfg.downsample_ratio, m = cfg.mask_margin_ratio;
  for (int i = 0; i < n && (int)rects.size() < kMaxRects; ++i) {
    const float cx = boxes[4 * i], cy = boxes[4 * i + 1], w = boxes[4 * i + 2] * (1.f + m), h = boxes[4 * i + 3] * (1.f + m);
    int x1 = (int)std::floor((cx - w / 2) * r), y1 = (int)std::floor((cy - h / 2) * r);
    int x2 = (int)std::ceil((cx + w / 2) * r), y2 = (int)std::ceil((cy + h / 2) * r);
    x1 = std::max(x1, 0); y1 = std::max(y1, 0); x2 = std::min(x2, gw - 1); y2 = std::min(y2, gh - 1);
    if (x2 >= x1 && y2 >= y1) rects.push_back(make_int4(x1, y1, x2, y2));
  }
}

void Stabilizer::Impl::gray_from_frame(const uint8_t* frame, int h, int w) {
  GTX_CHECK(h == fh && w == fw, "stabilizer: frame is %dx%d, created for %dx%d", w, h, fw, fh);
  const size_t bytes = (size_t)h * w * 3;
  if (d_frame.bytes < bytes) d_frame.alloc(bytes);
  GTX_HIP(hipMemcpyAsync(d_frame.p, frame, bytes, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(gray_kernel, dim3(cdiv(gw, 256), gh), dim3(256), 0, ctx->stream, d_frame.as<uint8_t>(), h, w, half,
                     d_pyr.as<uint8_t>(), gh, gw);
  GTX_HIP(hipGetLastError());
}

// gray_dev == nullptr: level 0 is already in d_pyr.
void Stabilizer::Impl::extract(const uint8_t* gray_dev, const float* boxes, int n, const Levels& Lplan, int slots, Feat& out) {
  hipStream_t s = ctx->stream;
  uint8_t* pyr = d_pyr.as<uint8_t>();
  Levels L = Lplan;   // level 0 is read in place from the caller's gray image when there is one
  for (int i = 0; i < L.n; ++i) L.l[i].img = pyr + L.l[i].off;
  if (gray_dev) L.l[0].img = gray_dev;
  if (cfg.clahe) {        // equalised level 0 goes into the pyramid buffer (the caller's gray image is shared with others)
    clahe_dev(L.l[0].img, gh, gw, d_clahe_lut.as<uint8_t>(), pyr, s);
    L.l[0].img = pyr;
  }
  for (int g = 0; g < L.n_groups; ++g) {
    const int first = L.grp_first[g], n = L.grp_n[g];
    if (n == 1) {                                   // a single level: the plain per-level kernel (4 pixels per thread, dword stores)
      const unsigned* tx = d_pyr_tab.as<unsigned>() + L.tab_off[first];
      hipLaunchKernelGGL(pyr_resize_kernel, dim3(cdiv(L.l[first].w, 1024), L.l[first].h), dim3(256), 0, s, L.l[first - 1].img, L.l[first - 1].w,
                         pyr + L.l[first].off, L.l[first].w, tx, tx + L.l[first].w);
      continue;
    }
    PyrGroup G{};
    G.n = n;
    G.src = L.l[first - 1].img;
    G.sw = L.l[first - 1].w;
    G.sh = L.l[first - 1].h;
    for (int k = 0; k < n; ++k) {
      G.dst[k] = pyr + L.l[first + k].off;
      G.w[k] = L.l[first + k].w;
      G.h[k] = L.l[first + k].h;
      G.tx[k] = d_pyr_tab.as<unsigned>() + L.tab_off[first + k];
    }
    G.buf_bytes = L.grp_buf[g];
    const Level& ll = L.l[first + n - 1];
    hipLaunchKernelGGL(pyr_group_kernel, dim3(cdiv(ll.w, 64), cdiv(ll.h, 16)), dim3(256), 2 * (size_t)G.buf_bytes, s, G);
  }
  int n_rects = 0;
  if (cfg.mask_use && boxes && n > 0) {
    std::vector<int4> rects;
    build_rects(boxes, n, rects);
    if (!rects.empty()) {
      GTX_HIP(hipMemcpyAsync(d_rects.p, rects.data(), sizeof(int4) * rects.size(), hipMemcpyHostToDevice, s));
      n_rects = (int)rects.size();
    }
  }
  // counters: [cand_n 8 x 16 sub-lists][elig_n 8][score histogram 8 x 256]; zero here: describe_kernel clears them at the end of every pass
  int* cand_n = d_counters.as<int>();
  int* elig_n = cand_n + kPyrLevels * kCandSub;
  int* hist = elig_n + kPyrLevels;
  hipLaunchKernelGGL(fast_detect_kernel, dim3(L.n_tiles), dim3(256), 0, s, d_rects.as<int4>(), n_rects, gw, gh, L, cfg.fast_threshold,
                     d_cand.as<Cand>(), cand_n, hist);
  hipLaunchKernelGGL(harris_kernel, dim3(512), dim3(256), 0, s, L, d_cand.as<Cand>(), cand_n, hist, d_elig.as<Cand>(), elig_n);
  hipLaunchKernelGGL(select_kernel, dim3(L.n), dim3(1024), kSortCap * 12, s, d_elig.as<Cand>(), elig_n, L, d_kps.as<KeyPoint>(),
                     d_kp_n.as<int>());
  hipLaunchKernelGGL(describe_kernel, dim3(cdiv(slots, 4)), dim3(256), 0, s, L, d_kps.as<KeyPoint>(), d_kp_n.as<int>(),
                     d_pattern.as<int8_t>(), out.kps.as<KeyPoint>(), out.desc.as<unsigned long long>(), out.xy.as<float2>(), out.n.as<int>(),
                     1.0f / cfg.downsample_ratio, slots, d_counters.as<int>(), kCounterInts);
  GTX_HIP(hipGetLastError());
}

namespace {

// Robust refinement of the RANSAC winner. Keypoints sit on integer pixels of their pyramid level,
// so matches carry 1-2 px of localisation noise in full-resolution pixels -- comparable to the
// RANSAC threshold. A hard inlier cut at the threshold throws away (and biases) much of the
// evidence. Instead: keep every match within 3x the threshold of the winner and minimise the
// geometric transfer error with iteratively re-weighted Gauss-Newton (Tukey biweight, scale from
// the median residual), h33 = 1, in normalised coordinates.
bool refine_homography(const std::vector<float4>& pts, double cx, double cy, double sc,
                       double thr, double H[9], int* n_inliers, bool affine = false) {
  const size_t n = pts.size();
  const int np = affine ? 6 : 8;                     // affine: h31 = h32 = 0 stay as they are, six parameters move
  const size_t min_pts = affine ? 3 : 4;
  // to normalised coordinates: Hn = T H T^-1
  auto to_norm = [&](const double* Hp, double* Hn) {
    const double is = 1.0 / sc;
    double M[9];  // M = H T^-1, T^-1 = [[is,0,cx],[0,is,cy],[0,0,1]]
    for (int r = 0; r < 3; ++r) {
      M[r * 3 + 0] = Hp[r * 3 + 0] * is;
      M[r * 3 + 1] = Hp[r * 3 + 1] * is;
      M[r * 3 + 2] = Hp[r * 3 + 0] * cx + Hp[r * 3 + 1] * cy + Hp[r * 3 + 2];
    }
    for (int k = 0; k < 3; ++k) {  // Hn = T M, T = [[sc,0,-sc cx],[0,sc,-sc cy],[0,0,1]]
      Hn[0 + k] = sc * M[0 + k] - sc * cx * M[6 + k];
      Hn[3 + k] = sc * M[3 + k] - sc * cy * M[6 + k];
      Hn[6 + k] = M[6 + k];
    }
    const double inv = 1.0 / Hn[8];
    for (int i = 0; i < 9; ++i) Hn[i] *= inv;
  };
  auto from_norm = [&](const double* Hn, double* Hp) {
    const double is = 1.0 / sc;
    double M[9];  // M = Hn T
    for (int r = 0; r < 3; ++r) {
      M[r * 3 + 0] = Hn[r * 3 + 0] * sc;
      M[r * 3 + 1] = Hn[r * 3 + 1] * sc;
      M[r * 3 + 2] = Hn[r * 3 + 2] - sc * (Hn[r * 3 + 0] * cx + Hn[r * 3 + 1] * cy);
    }
    for (int k = 0; k < 3; ++k) {
      Hp[0 + k] = is * M[0 + k] + cx * M[6 + k];
      Hp[3 + k] = is * M[3 + k] + cy * M[6 + k];
      Hp[6 + k] = M[6 + k];
    }
    const double inv = 1.0 / Hp[8];
    for (int i = 0; i < 9; ++i) Hp[i] *= inv;
  };
  std::vector<double> x(n), y(n), u(n), v(n);
  for (size_t i = 0; i < n; ++i) {
    x[i] = (pts[i].x - cx) * sc; y[i] = (pts[i].y - cy) * sc;
    u[i] = (pts[i].z - cx) * sc; v[i] = (pts[i].w - cy) * sc;
  }
  double h[9];
  to_norm(H, h);
  auto residual = [&](size_t i, double& rx, double& ry, double& w) {
    w = h[6] * x[i] + h[7] * y[i] + 1.0;
    rx = (h[0] * x[i] + h[1] * y[i] + h[2]) / w - u[i];
    ry = (h[3] * x[i] + h[4] * y[i] + h[5]) / w - v[i];
  };
  // support set: within 3 thresholds of the winner (fixed over the iterations)
  std::vector<int> sup;
  const double lim = 3.0 * thr * sc;
  for (size_t i = 0; i < n; ++i) {
    double rx, ry, w;
    residual(i, rx, ry, w);
    if (std::fabs(w) > 1e-9 && rx * rx + ry * ry <= lim * lim) sup.push_back((int)i);
  }
  if (sup.size() < min_pts) return false;
  std::vector<double> un(sup.size());
  for (int iter = 0; iter < 8; ++iter) {
    for (size_t k = 0; k < sup.size(); ++k) {
      double rx, ry, w;
      residual(sup[k], rx, ry, w);
      un[k] = std::sqrt(rx * rx + ry * ry) / sc;   // pixels
    }
    std::vector<double> tmp = un;
    std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
    const double sigma = std::max(1.4826 * tmp[tmp.size() / 2], 0.05);
    const double c = 4.685 * sigma;
    double A[64] = {0}, g[8] = {0};
    for (size_t k = 0; k < sup.size(); ++k) {
      const int i = sup[k];
      if (un[k] >= c) continue;
      const double t = 1.0 - (un[k] / c) * (un[k] / c);
      const double wt = t * t;
      double rx, ry, w;
      residual(i, rx, ry, w);
      const double iw = 1.0 / w, px = rx + u[i], py = ry + v[i];
      const double Jx[8] = {x[i] * iw, y[i] * iw, iw, 0, 0, 0, -px * x[i] * iw, -px * y[i] * iw};
      const double Jy[8] = {0, 0, 0, x[i] * iw, y[i] * iw, iw, -py * x[i] * iw, -py * y[i] * iw};
      for (int a = 0; a < np; ++a) {
        g[a] += wt * (Jx[a] * rx + Jy[a] * ry);
        for (int b = a; b < np; ++b) A[a * 8 + b] += wt * (Jx[a] * Jx[b] + Jy[a] * Jy[b]);
      }
    }
    for (int a = 0; a < np; ++a)
      for (int b = 0; b < a; ++b) A[a * 8 + b] = A[b * 8 + a];
    // Cholesky solve A d = g
    double Lc[64] = {0};
    bool ok = true;
    for (int i = 0; i < np && ok; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = A[i * 8 + j];
        for (int k = 0; k < j; ++k) s -= Lc[i * 8 + k] * Lc[j * 8 + k];
        if (i == j) {
          if (!(s > 0)) { ok = false; break; }
          Lc[i * 8 + i] = std::sqrt(s);
        } else {
          Lc[i * 8 + j] = s / Lc[j * 8 + j];
        }
      }
    if (!ok) break;
    double yv[8], d[8];
    for (int i = 0; i < np; ++i) {
      double s = g[i];
      for (int k = 0; k < i; ++k) s -= Lc[i * 8 + k] * yv[k];
      yv[i] = s / Lc[i * 8 + i];
    }
    for (int i = np - 1; i >= 0; --i) {
      double s = yv[i];
      for (int k = i + 1; k < np; ++k) s -= Lc[k * 8 + i] * d[k];
      d[i] = s / Lc[i * 8 + i];
    }
    double step = 0;
    for (int a = 0; a < np; ++a) { h[a] -= d[a]; step = std::max(step, std::fabs(d[a])); }
    if (step < 1e-14) break;
  }
  from_norm(h, H);
  int cnt = 0;
  const double t2 = thr * sc * thr * sc;
  for (size_t i = 0; i < n; ++i) {
    double rx, ry, w;
    residual(i, rx, ry, w);
    if (rx * rx + ry * ry <= t2) ++cnt;
  }
  if (n_inliers) *n_inliers = cnt;
  return cnt >= (int)min_pts;
}

}  // namespace

// Asynchronous half of a stabilize pass: match -> ratio -> RANSAC on the stream, then two D2H
// copies (result record, match points) into pinned memory and an event.
void Stabilizer::Impl::submit_match() {
  hipStream_t s = ctx->stream;
  const int max_q = slots_cur, n_chunks = cdiv(slots_ref, kMatchChunk);
  hipLaunchKernelGGL(match_kernel, dim3(cdiv(slots_cur, 256), n_chunks), dim3(256), 0, s, cur.desc.as<unsigned long long>(),
                     cur.n.as<int>(), ref.desc.as<unsigned long long>(), ref.n.as<int>(), max_q, d_pidx.as<int>(), d_pd1.as<int>(),
                     d_pd2.as<int>(), d_mticket.as<unsigned>(), cfg.filter_ratio, cfg.filter_type == 1 ? 1 : 0, cur.xy.as<float2>(), ref.xy.as<float2>(),
                     d_bidx.as<int>(), d_bd.as<int>(), d_sd.as<int>(), d_mq.as<int>(), d_mt.as<int>(), d_md.as<int>(), d_mpts(),
                     d_nmatch.as<int>());
  const double cx = fw / 2.0, cy = fh / 2.0, sc = 2.0 / fw;
  const float thr2 = cfg.ransac_threshold * cfg.ransac_threshold;
  hipLaunchKernelGGL(ransac_kernel, dim3(cdiv(n_hyp, 8)), dim3(512), 0, s, d_mpts(), d_nmatch.as<int>(), cur.n.as<int>(), cfg.seed, n_hyp,
                     cx, cy, sc, cfg.affine ? 1 : 0, thr2, d_rstate.as<unsigned long long>(), d_res());
  GTX_HIP(hipGetLastError());
  GTX_HIP(hipMemcpyAsync(h_out, d_out.p, kOutPts + sizeof(float4) * slots_cur, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipEventRecord(t1_ev, s));
  GTX_HIP(hipEventRecord(done_ev, s));
  pending = true;
}

// Blocking half: wait for the pass, then the robust refit on the host (f64).
void Stabilizer::Impl::collect(double Hout[9], int* valid_out, int st[4]) {
  GTX_CHECK(pending, "stabilizer: collect without a submitted frame");
  GTX_HIP(hipEventSynchronize(done_ev));
  pending = false;
  last_ms = 0.f;
  if (timed) GTX_HIP(hipEventElapsedTime(&last_ms, t0_ev, t1_ev));
  timed = false;
  const StabResult& R = *h_res;
  const double cx = fw / 2.0, cy = fh / 2.0, sc = 2.0 / fw;
  cur.host_n = R.n_cur;
  stats[0] = ref.host_n; stats[1] = R.n_cur; stats[2] = R.n_match; stats[3] = 0;
  valid = false;
  if (R.n_match >= (cfg.affine ? 3 : 4) && R.best >= 0) {
    std::vector<float4> pts(h_pts, h_pts + R.n_match);
    double Hc[9];
    std::memcpy(Hc, R.H, sizeof Hc);
    const double inv = 1.0 / Hc[8];
    for (double& v : Hc) v *= inv;
    int n_inl = 0;
    if (refine_homography(pts, cx, cy, sc, (double)cfg.ransac_threshold, Hc, &n_inl, cfg.affine != 0)) {
      std::memcpy(H, Hc, sizeof H);
      valid = true;
      stats[3] = n_inl;
    }
  }
  if (Hout) std::memcpy(Hout, H, sizeof H);
  if (valid_out) *valid_out = valid ? 1 : 0;
  if (st) std::memcpy(st, stats, sizeof stats);
}

void Stabilizer::set_ref_frame(const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n) {
  Impl& S = *impl_;
  GTX_HIP(hipSetDevice(S.ctx->device));
  S.gray_from_frame(frame_bgr, h, w);
  S.extract(nullptr, boxes_xywh, n, S.lev_ref, S.slots_ref, S.ref);
  GTX_HIP(hipMemcpyAsync(&S.ref.host_n, S.ref.n.p, sizeof(int), hipMemcpyDeviceToHost, S.ctx->stream));
  GTX_HIP(hipStreamSynchronize(S.ctx->stream));
  S.have_ref = true;
}

void Stabilizer::set_ref_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n) {
  Impl& S = *impl_;
  GTX_CHECK(gh == S.gh && gw == S.gw, "stabilizer: gray image is %dx%d, expected %dx%d", gw, gh, S.gw, S.gh);
  GTX_HIP(hipSetDevice(S.ctx->device));
  S.extract(static_cast<const uint8_t*>(gray), boxes_xywh, n, S.lev_ref, S.slots_ref, S.ref);
  GTX_HIP(hipMemcpyAsync(&S.ref.host_n, S.ref.n.p, sizeof(int), hipMemcpyDeviceToHost, S.ctx->stream));
  GTX_HIP(hipStreamSynchronize(S.ctx->stream));
  S.have_ref = true;
}

void Stabilizer::stabilize(const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n, double H[9], int* valid, int stats[4]) {
  Impl& S = *impl_;
  if (!S.have_ref) fail(GTX_ERR_STATE, "stabilize called before set_ref_frame");
  GTX_HIP(hipSetDevice(S.ctx->device));
  S.gray_from_frame(frame_bgr, h, w);
  S.extract(nullptr, boxes_xywh, n, S.lev_cur, S.slots_cur, S.cur);
  S.submit_match();
  S.collect(H, valid, stats);
}

void Stabilizer::submit_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n) {
  Impl& S = *impl_;
  if (!S.have_ref) fail(GTX_ERR_STATE, "stabilize called before set_ref_frame");
  GTX_CHECK(!S.pending, "stabilizer: a frame is already in flight");
  GTX_CHECK(gh == S.gh && gw == S.gw, "stabilizer: gray image is %dx%d, expected %dx%d", gw, gh, S.gw, S.gh);
  GTX_HIP(hipSetDevice(S.ctx->device));
  GTX_HIP(hipEventRecord(S.t0_ev, S.ctx->stream));
  S.timed = true;
  S.extract(static_cast<const uint8_t*>(gray), boxes_xywh, n, S.lev_cur, S.slots_cur, S.cur);
  S.submit_match();
}

// The features of the frame stabilized last become the reference (their buffers are swapped): a caller that registers every
// frame against the one before it (gmc.FeatureGMC) does not extract each frame's features twice. Both plans must be the same.
void Stabilizer::promote_cur_to_ref() {
  Impl& S = *impl_;
  if (!S.have_ref) fail(GTX_ERR_STATE, "promote_cur_to_ref before a frame was stabilized");
  GTX_CHECK(!S.pending, "stabilizer: a frame is in flight");
  GTX_CHECK(S.slots_ref == S.slots_cur, "stabilizer: the reference holds %d keypoint slots, a frame %d (ref_multiplier must be 1)", S.slots_ref, S.slots_cur);
  std::swap(S.ref, S.cur);
}

float Stabilizer::last_ms() const { return impl_->last_ms; }

void Stabilizer::collect(double H[9], int* valid, int stats[4]) {
  GTX_HIP(hipSetDevice(impl_->ctx->device));
  impl_->collect(H, valid, stats);
}

void Stabilizer::stabilize_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n, double H[9], int* valid, int stats[4]) {
  submit_gray_dev(gray, gh, gw, boxes_xywh, n);
  collect(H, valid, stats);
}

void Stabilizer::keypoints(int which, int cap, int* n, float* xy, int* level, int* angle_bin, uint8_t* desc) {
  Impl& S = *impl_;
  Impl::Feat& f = which == 0 ? S.ref : S.cur;
  GTX_HIP(hipSetDevice(S.ctx->device));
  GTX_HIP(hipStreamSynchronize(S.ctx->stream));
  const int k = std::min(cap, f.host_n);
  *n = k;
  if (k == 0) return;
  std::vector<KeyPoint> kp(k);
  GTX_HIP(hipMemcpy(kp.data(), f.kps.p, sizeof(KeyPoint) * k, hipMemcpyDeviceToHost));
  if (xy) GTX_HIP(hipMemcpy(xy, f.xy.p, sizeof(float2) * k, hipMemcpyDeviceToHost));
  if (desc) GTX_HIP(hipMemcpy(desc, f.desc.p, 32 * (size_t)k, hipMemcpyDeviceToHost));
  for (int i = 0; i < k; ++i) {
    if (level) level[i] = kp[i].level;
    if (angle_bin) angle_bin[i] = kp[i].bin;
  }
}

void Stabilizer::matches(int cap, int* n, int* cur_idx, int* ref_idx, int* dist) {
  Impl& S = *impl_;
  GTX_HIP(hipSetDevice(S.ctx->device));
  GTX_HIP(hipStreamSynchronize(S.ctx->stream));
  const int k = std::min(cap, S.stats[2]);
  *n = k;
  if (k == 0) return;
  if (cur_idx) GTX_HIP(hipMemcpy(cur_idx, S.d_mq.p, 4 * (size_t)k, hipMemcpyDeviceToHost));
  if (ref_idx) GTX_HIP(hipMemcpy(ref_idx, S.d_mt.p, 4 * (size_t)k, hipMemcpyDeviceToHost));
  if (dist) GTX_HIP(hipMemcpy(dist, S.d_md.p, 4 * (size_t)k, hipMemcpyDeviceToHost));
}

void Stabilizer::pattern(int8_t* out) const { std::memcpy(out, impl_->pattern.data(), impl_->pattern.size()); }

// Blocking robust homography from matched point pairs already in HBM (registration path): the same
// hypothesis / MSAC-score / argmin kernels as the per-frame stabilizer, then the host IRLS refit.
// cv2.createCLAHE(2.0, (8, 8)).apply(gray) for a host image (the stabilizer's own pre-processing step, exposed for tests / tools).
void clahe_image(gtx_ctx* ctx, const uint8_t* gray, int h, int w, uint8_t* out) {
  GTX_HIP(hipSetDevice(ctx->device));
  GTX_CHECK(h >= 8 && w >= 8, "clahe: image %dx%d is smaller than the tile grid", w, h);
  const size_t bytes = (size_t)h * w;
  DevBuf d_src(bytes), d_dst(bytes), d_lut(kClaheLutBytes);
  hipStream_t s = ctx->stream;
  GTX_HIP(hipMemcpyAsync(d_src.p, gray, bytes, hipMemcpyHostToDevice, s));
  clahe_dev(d_src.as<uint8_t>(), h, w, d_lut.as<uint8_t>(), d_dst.as<uint8_t>(), s);
  GTX_HIP(hipGetLastError());
  GTX_HIP(hipMemcpyAsync(out, d_dst.p, bytes, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
}

bool ransac_homography(int device, hipStream_t s, const float4* d_pts, int n_match, unsigned seed, int n_hyp, int frame_w, int frame_h,
                       float threshold, double H[9], int* n_inliers) {
  *n_inliers = 0;
  if (n_match < 4) return false;
  GTX_HIP(hipSetDevice(device));
  DevBuf d_n(sizeof(int) * 2), d_state(2 * sizeof(unsigned long long)), d_res(sizeof(StabResult));
  const int two[2] = {n_match, n_match};
  const unsigned long long armed[2] = {kNoHyp, 0ull};
  GTX_HIP(hipMemcpyAsync(d_n.p, two, sizeof two, hipMemcpyHostToDevice, s));
  GTX_HIP(hipMemcpyAsync(d_state.p, armed, sizeof armed, hipMemcpyHostToDevice, s));
  const double cx = frame_w / 2.0, cy = frame_h / 2.0, sc = 2.0 / frame_w;
  GTX_CHECK(n_hyp <= 65536, "ransac: at most 65536 hypotheses (got %d)", n_hyp);
  hipLaunchKernelGGL(ransac_kernel, dim3(cdiv(n_hyp, 8)), dim3(512), 0, s, d_pts, d_n.as<int>(), d_n.as<int>() + 1, seed, n_hyp, cx, cy, sc, 0,
                     threshold * threshold, d_state.as<unsigned long long>(), d_res.as<StabResult>());
  GTX_HIP(hipGetLastError());
  StabResult R;
  std::vector<float4> pts(n_match);
  GTX_HIP(hipMemcpyAsync(&R, d_res.p, sizeof R, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipMemcpyAsync(pts.data(), d_pts, sizeof(float4) * n_match, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
  if (R.best < 0) return false;
  double Hc[9];
  std::memcpy(Hc, R.H, sizeof Hc);
  const double inv = 1.0 / Hc[8];
  for (double& v : Hc) v *= inv;
  if (!refine_homography(pts, cx, cy, sc, (double)threshold, Hc, n_inliers)) return false;
  std::memcpy(H, Hc, sizeof Hc);
  return true;
}

}  // namespace gtx
