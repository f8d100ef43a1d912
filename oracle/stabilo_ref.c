/* ORACLE -- test infrastructure, not product code.
 *
 * Plain-C restatement of oracle/stabilo_ref.py (which restates what stabilo.Stabilizer computes for the reference:
 * call sites geotrax/extract.py:139,177-187, parameters geotrax/cfg/default.yaml:100-145), function by function, in the
 * same integer-exact form: gray conversion, integer bilinear pyramid, FAST-9/16 + 3x3 NMS + foreground mask, FAST-score
 * pre-selection then Harris ranking, intensity-centroid orientation, steered BRIEF on the Gaussian-smoothed patch,
 * Hamming 2-NN + Lowe ratio, RANSAC (MSAC score) + iteratively re-weighted refit. It exists for ONE purpose: the CPU
 * baseline of bench.py (BASELINE.md section 2.1 names a "C++ ORB + Hamming matcher + RANSAC" timed on 1 thread and on
 * all host cores; the numpy oracle is single-threaded and spends its time in Python loops). tests/test_stabilo_c.py holds it
 * against the numpy oracle: every integer stage bit for bit, the homography to 1e-6 of a pixel on a 9 x 16 grid.
 * PARITY UNPINNED against stabilo / OpenCV themselves, exactly like the file it restates.
 *
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may build or load this.
 *   gcc -O2 -fopenmp -shared -fPIC -o oracle/_build/libstabilo_ref.so oracle/stabilo_ref.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BORDER 31
#define N_BINS 256
static const int CIRCLE[16][2] = {{0, -3}, {1, -3}, {2, -2}, {3, -1}, {3, 0}, {3, 1}, {2, 2}, {1, 3},
                                  {0, 3}, {-1, 3}, {-2, 2}, {-3, 1}, {-3, 0}, {-3, -1}, {-2, -2}, {-1, -3}};
static const int UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
static const int64_t GAUSS[7] = {18, 34, 49, 54, 49, 34, 18};

void stab_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

/* stabilo_ref.py:88-93 */
void stab_gray(const uint8_t* bgr, int h, int w, int half, uint8_t* out) {
  if (!half) {
#pragma omp parallel for
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const uint8_t* p = bgr + ((size_t)y * w + x) * 3;
        out[(size_t)y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
      }
    return;
  }
  const int oh = h / 2, ow = w / 2;
#pragma omp parallel for
  for (int y = 0; y < oh; ++y)
    for (int x = 0; x < ow; ++x) {
      int s = 0;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const uint8_t* p = bgr + ((size_t)(2 * y + dy) * w + 2 * x + dx) * 3;
          s += (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14;
        }
      out[(size_t)y * ow + x] = (uint8_t)((s + 2) >> 2);
    }
}

/* stabilo_ref.py:96-113 */
static void resize_int(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  int* x0 = malloc(sizeof(int) * dw * 3);
  int *x1 = x0 + dw, *fx = x1 + dw;
  for (int d = 0; d < dw; ++d) {
    int64_t fp = ((int64_t)(2 * d + 1) * sw * 32768) / dw - 32768;
    if (fp < 0) fp = 0;
    const int i0 = (int)(fp >> 16);
    x0[d] = i0 < sw - 1 ? i0 : sw - 1;
    x1[d] = i0 + 1 < sw - 1 ? i0 + 1 : sw - 1;
    fx[d] = (int)((fp >> 5) & 2047);
  }
#pragma omp parallel for
  for (int y = 0; y < dh; ++y) {
    int64_t fp = ((int64_t)(2 * y + 1) * sh * 32768) / dh - 32768;
    if (fp < 0) fp = 0;
    const int i0 = (int)(fp >> 16);
    const int y0 = i0 < sh - 1 ? i0 : sh - 1, y1 = i0 + 1 < sh - 1 ? i0 + 1 : sh - 1;
    const int64_t fy = (fp >> 5) & 2047;
    const uint8_t *r0 = src + (size_t)y0 * sw, *r1 = src + (size_t)y1 * sw;
    for (int x = 0; x < dw; ++x) {
      const int64_t top = (int64_t)r0[x0[x]] * (2048 - fx[x]) + (int64_t)r0[x1[x]] * fx[x];
      const int64_t bot = (int64_t)r1[x0[x]] * (2048 - fx[x]) + (int64_t)r1[x1[x]] * fx[x];
      dst[(size_t)y * dw + x] = (uint8_t)((top * (2048 - fy) + bot * fy + (1 << 21)) >> 22);
    }
  }
  free(x0);
}

/* stabilo_ref.py:134-150, one pixel */
static int fast_score_px(const uint8_t* c, int w, int thr) {
  const int p = c[0];
  int d[16];
  for (int i = 0; i < 16; ++i) d[i] = (int)c[CIRCLE[i][1] * w + CIRCLE[i][0]] - p;
  int best = 0;
  for (int k = 0; k < 16; ++k) {
    int mb = d[k], md = -d[k];
    for (int j = 1; j < 9; ++j) {
      const int v = d[(k + j) & 15];
      if (v < mb) mb = v;
      if (-v < md) md = -v;
    }
    if (mb > best) best = mb;
    if (md > best) best = md;
  }
  return best > thr ? (best < 255 ? best : 255) : 0;
}

/* stabilo_ref.py:153-167 */
static int64_t harris_key(const uint8_t* img, int w, int y0, int x0) {
  int64_t a = 0, b = 0, c = 0;
  for (int dy = -3; dy <= 3; ++dy)
    for (int dx = -3; dx <= 3; ++dx) {
      const uint8_t* q = img + (size_t)(y0 + dy) * w + x0 + dx;
      const int64_t ix = ((int64_t)q[-w + 1] + 2 * q[1] + q[w + 1]) - ((int64_t)q[-w - 1] + 2 * q[-1] + q[w - 1]);
      const int64_t iy = ((int64_t)q[w - 1] + 2 * q[w] + q[w + 1]) - ((int64_t)q[-w - 1] + 2 * q[-w] + q[-w + 1]);
      a += ix * ix; b += iy * iy; c += ix * iy;
    }
  return 25 * (a * b - c * c) - (a + b) * (a + b);
}

/* stabilo_ref.py:170-181; tan table built like TAN there */
static int64_t TAN[32];
static int tan_ready = 0;
static int angle_bin(int64_t m10, int64_t m01) {
  if (!tan_ready) {
    for (int j = 0; j < 32; ++j) TAN[j] = (int64_t)floor(tan((j + 0.5) * 2.0 * M_PI / N_BINS) * 16777216.0 + 0.5);
    tan_ready = 1;
  }
  const int64_t ax = m10 < 0 ? -m10 : m10, ay = m01 < 0 ? -m01 : m01;
  const int swap = ay > ax;
  const int64_t hi = swap ? ay : ax, lo = swap ? ax : ay;
  int o = 0;
  if (hi > 0)
    for (int j = 0; j < 32; ++j) o += ((lo << 24) >= hi * TAN[j]) ? 1 : 0;
  if (swap) o = 64 - o;
  if (m10 < 0) o = 128 - o;
  if (m01 < 0) o = -o;
  return o & (N_BINS - 1);
}

typedef struct { int64_t key; int pix; int score; } Cand;
static int cmp_score_desc(const void* a, const void* b) { return ((const Cand*)b)->score - ((const Cand*)a)->score; }
static int cmp_key_desc_pix_asc(const void* a, const void* b) {
  const Cand *x = a, *y = b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  return x->pix < y->pix ? -1 : (x->pix > y->pix ? 1 : 0);
}

/* stabilo_ref.py:116-131 */
static void level_plan(int gw, int gh, int n_levels, float scale_factor, int max_features, int* w, int* h, float* sc, int* n_want) {
  const double factor = 1.0 / (double)scale_factor;
  double want = max_features * (1.0 - factor) / (1.0 - pow(factor, n_levels));
  int total = 0;
  for (int i = 0; i < n_levels; ++i) {
    const double s = pow((double)scale_factor, i);
    w[i] = (int)floor(gw / s + 0.5);
    h[i] = (int)floor(gh / s + 0.5);
    sc[i] = (float)s;
    if (i < n_levels - 1) {
      n_want[i] = (int)floor(want + 0.5);
      total += n_want[i];
      want *= factor;
    } else {
      n_want[i] = max_features - total > 0 ? max_features - total : 0;
    }
  }
}

/* stabilo_ref.py:197-263. gray: level 0 [gh][gw]; rects: n_rects x (x1, y1, x2, y2) inclusive, level-0 pixels (mask_rects is
 * host arithmetic and stays in Python); pattern [256][256][4] int8 (brief_pattern(), generated in Python).
 * Outputs (cap rows): xy [K][2] f32, level, bin, desc [K][32], px [K][2]. Returns K. */
int stab_extract(const uint8_t* gray, int gh, int gw, const int* rects, int n_rects, int n_levels, float scale_factor, int fast_threshold,
                 float downsample_ratio, int max_features, const int8_t* pattern, int cap, float* xy, int* level, int* bin, uint8_t* desc, int* px) {
  int w[16], h[16], n_want[16];
  float sc[16];
  if (n_levels > 16) return -1;
  level_plan(gw, gh, n_levels, scale_factor, max_features, w, h, sc, n_want);
  uint8_t* mask = NULL;
  if (n_rects > 0) {
    mask = malloc((size_t)gh * gw);
    memset(mask, 255, (size_t)gh * gw);
    for (int r = 0; r < n_rects; ++r)
      for (int y = rects[4 * r + 1]; y <= rects[4 * r + 3]; ++y) memset(mask + (size_t)y * gw + rects[4 * r], 0, (size_t)(rects[4 * r + 2] - rects[4 * r] + 1));
  }
  const float inv_ratio = 1.0f / downsample_ratio;
  const uint8_t* img = gray;
  uint8_t* owned = NULL;
  int K = 0;
  for (int li = 0; li < n_levels; ++li) {
    const int W = w[li], H = h[li];
    if (li > 0) {
      uint8_t* next = malloc((size_t)W * H);
      resize_int(img, w[li - 1], h[li - 1], next, W, H);
      free(owned);
      owned = next;
      img = next;
    }
    if (H <= 2 * BORDER || W <= 2 * BORDER) continue;
    uint8_t* score = calloc((size_t)W * H, 1);
#pragma omp parallel for schedule(dynamic, 8)
    for (int y = BORDER; y < H - BORDER; ++y)
      for (int x = BORDER; x < W - BORDER; ++x) score[(size_t)y * W + x] = (uint8_t)fast_score_px(img + (size_t)y * W + x, W, fast_threshold);
    /* 3x3 strict maxima, foreground test */
    Cand* cand = malloc(sizeof(Cand) * (size_t)(W / 2 + 1) * (size_t)(H / 2 + 1));
    int n = 0;
    for (int y = 1; y < H - 1; ++y)
      for (int x = 1; x < W - 1; ++x) {
        const uint8_t* s = score + (size_t)y * W + x;
        const int v = s[0];
        if (v == 0) continue;
        if (!(v > s[-W - 1] && v > s[-W] && v > s[-W + 1] && v > s[-1] && v > s[1] && v > s[W - 1] && v > s[W] && v > s[W + 1])) continue;
        if (mask) {
          int64_t x0 = ((int64_t)x * gw + W / 2) / W, y0 = ((int64_t)y * gh + H / 2) / H;
          if (x0 > gw - 1) x0 = gw - 1;
          if (y0 > gh - 1) y0 = gh - 1;
          if (mask[(size_t)y0 * gw + x0] == 0) continue;
        }
        cand[n].pix = y * W + x; cand[n].score = v; cand[n].key = 0; ++n;
      }
    free(score);
    if (n == 0 || n_want[li] == 0) { free(cand); continue; }
    /* stage 1: the 2 n_want best FAST scores, everything tied with the last kept */
    if (n > 2 * n_want[li]) {
      int hist[256] = {0};
      for (int i = 0; i < n; ++i) hist[cand[i].score]++;
      int cut = 255, acc = 0;
      for (; cut > 0; --cut) { acc += hist[cut]; if (acc >= 2 * n_want[li]) break; }
      int m = 0;
      for (int i = 0; i < n; ++i) if (cand[i].score >= cut) cand[m++] = cand[i];
      n = m;
    }
    /* stage 2: the n_want best Harris responses, ties to the smaller pixel index */
#pragma omp parallel for
    for (int i = 0; i < n; ++i) cand[i].key = harris_key(img, W, cand[i].pix / W, cand[i].pix % W);
    qsort(cand, (size_t)n, sizeof(Cand), cmp_key_desc_pix_asc);
    const int keep = n < n_want[li] ? n : n_want[li];
    if (K + keep > cap) { free(cand); free(owned); free(mask); return -2; }
#pragma omp parallel for
    for (int i = 0; i < keep; ++i) {
      const int y = cand[i].pix / W, x = cand[i].pix % W;
      const uint8_t* c = img + (size_t)y * W + x;
      int64_t m10 = 0, m01 = 0;
      for (int v = -15; v <= 15; ++v) {
        const int u = UMAX[v < 0 ? -v : v];
        int64_t rs = 0;
        for (int t = -u; t <= u; ++t) { const int val = c[v * W + t]; m10 += (int64_t)t * val; rs += val; }
        m01 += (int64_t)v * rs;
      }
      const int b = angle_bin(m10, m01);
      int64_t hp[41][35];
      for (int r = 0; r < 41; ++r)
        for (int cc = 0; cc < 35; ++cc) {
          int64_t acc = 0;
          for (int k = 0; k < 7; ++k) acc += GAUSS[k] * c[(r - 20) * W + (cc + k - 20)];
          hp[r][cc] = acc;
        }
      uint8_t blur[35][35];
      for (int r = 0; r < 35; ++r)
        for (int cc = 0; cc < 35; ++cc) {
          int64_t acc = 0;
          for (int k = 0; k < 7; ++k) acc += GAUSS[k] * hp[r + k][cc];
          blur[r][cc] = (uint8_t)((acc + 32768) >> 16);
        }
      const int8_t* pt = pattern + (size_t)b * 256 * 4;
      uint8_t* d = desc + (size_t)(K + i) * 32;
      memset(d, 0, 32);
      for (int t = 0; t < 256; ++t) {
        const int va = blur[pt[4 * t + 1] + 17][pt[4 * t] + 17], vb = blur[pt[4 * t + 3] + 17][pt[4 * t + 2] + 17];
        if (va < vb) d[t >> 3] |= (uint8_t)(1u << (t & 7));   /* bit t of the 256-bit descriptor, little endian */
      }
      bin[K + i] = b;
      level[K + i] = li;
      px[2 * (K + i)] = x; px[2 * (K + i) + 1] = y;
      xy[2 * (K + i)] = (float)x * sc[li] * inv_ratio;
      xy[2 * (K + i) + 1] = (float)y * sc[li] * inv_ratio;
    }
    K += keep;
    free(cand);
  }
  free(owned);
  free(mask);
  return K;
}

/* stabilo_ref.py:269-287 */
int stab_match(const uint8_t* dq, int nq, const uint8_t* dt, int nt, float ratio, int keep_all, int* qi, int* ti, int* di) {
  if (nq == 0 || nt < (keep_all ? 1 : 2)) return 0;
  int* b1 = malloc(sizeof(int) * nq * 3);
  int *b2 = b1 + nq, *bi = b2 + nq;
#pragma omp parallel for
  for (int i = 0; i < nq; ++i) {
    const uint64_t* a = (const uint64_t*)(dq + (size_t)i * 32);
    int d1 = 1 << 30, d2 = 1 << 30, best = -1;
    for (int j = 0; j < nt; ++j) {
      const uint64_t* t = (const uint64_t*)(dt + (size_t)j * 32);
      const int d = __builtin_popcountll(a[0] ^ t[0]) + __builtin_popcountll(a[1] ^ t[1]) + __builtin_popcountll(a[2] ^ t[2]) + __builtin_popcountll(a[3] ^ t[3]);
      if (d < d1) { d2 = d1; d1 = d; best = j; }
      else if (d < d2) d2 = d;
    }
    b1[i] = d1; b2[i] = d2; bi[i] = best;
  }
  int n = 0;
  for (int i = 0; i < nq; ++i)
    if (keep_all || (float)b1[i] < ratio * (float)b2[i]) { qi[n] = i; ti[n] = bi[i]; di[n] = b1[i]; ++n; }
  free(b1);
  return n;
}

/* ---- RANSAC (stabilo_ref.py:290-408) ---- */
static uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

/* Gaussian elimination with partial pivoting, n <= 8 (np.linalg.solve's algorithm; last bits may differ from LAPACK's) */
static int solve_n(int n, double* A /* n x n row major, destroyed */, double* b /* in: rhs, out: solution */) {
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r) if (fabs(A[r * n + c]) > fabs(A[p * n + c])) p = r;
    if (!(fabs(A[p * n + c]) > 1e-300)) return 0;
    if (p != c) {
      for (int k = 0; k < n; ++k) { const double t = A[c * n + k]; A[c * n + k] = A[p * n + k]; A[p * n + k] = t; }
      const double t = b[c]; b[c] = b[p]; b[p] = t;
    }
    for (int r = c + 1; r < n; ++r) {
      const double f = A[r * n + c] / A[c * n + c];
      for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = n - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * b[k];
    b[r] = s / A[r * n + r];
  }
  return 1;
}

static void mat3_mul(const double* A, const double* B, double* C) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
/* T = [[sc,0,-sc cx],[0,sc,-sc cy],[0,0,1]];  Ti = T^-1 */
static void norm_mats(double cx, double cy, double sc, double* T, double* Ti) {
  const double t[9] = {sc, 0, -sc * cx, 0, sc, -sc * cy, 0, 0, 1}, ti[9] = {1 / sc, 0, cx, 0, 1 / sc, cy, 0, 0, 1};
  memcpy(T, t, sizeof t); memcpy(Ti, ti, sizeof ti);
}

static int cmp_double(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return x < y ? -1 : (x > y ? 1 : 0); }

/* stabilo_ref.py:305-350 */
static int refine(const double* H0, const double* p, const double* q, int n, double cx, double cy, double sc, double thr, int affine, double* Hout, int* n_inl) {
  const int npar = affine ? 6 : 8, min_pts = affine ? 3 : 4;
  double T[9], Ti[9], tmp[9], h[9];
  norm_mats(cx, cy, sc, T, Ti);
  mat3_mul(T, H0, tmp); mat3_mul(tmp, Ti, h);
  for (int i = 0; i < 9; ++i) tmp[i] = h[i] / h[8];
  memcpy(h, tmp, sizeof h);
  double* x = malloc(sizeof(double) * n * 8);
  double *y = x + n, *u = y + n, *v = u + n, *rx = v + n, *ry = rx + n, *w = ry + n, *un = w + n;
  int* sup = malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) { x[i] = (p[2 * i] - cx) * sc; y[i] = (p[2 * i + 1] - cy) * sc; u[i] = (q[2 * i] - cx) * sc; v[i] = (q[2 * i + 1] - cy) * sc; }
#define RES() for (int i = 0; i < n; ++i) { w[i] = h[6] * x[i] + h[7] * y[i] + 1.0; rx[i] = (h[0] * x[i] + h[1] * y[i] + h[2]) / w[i] - u[i]; ry[i] = (h[3] * x[i] + h[4] * y[i] + h[5]) / w[i] - v[i]; }
  RES()
  int ns = 0;
  const double lim2 = (3.0 * thr * sc) * (3.0 * thr * sc);
  for (int i = 0; i < n; ++i) if (fabs(w[i]) > 1e-9 && rx[i] * rx[i] + ry[i] * ry[i] <= lim2) sup[ns++] = i;
  int ok = 0;
  if (ns >= min_pts) {
    double* srt = malloc(sizeof(double) * ns);
    for (int it = 0; it < 8; ++it) {
      RES()
      for (int k = 0; k < ns; ++k) { const int i = sup[k]; un[k] = sqrt(rx[i] * rx[i] + ry[i] * ry[i]) / sc; srt[k] = un[k]; }
      qsort(srt, (size_t)ns, sizeof(double), cmp_double);
      const double sigma = fmax(1.4826 * srt[ns / 2], 0.05), c = 4.685 * sigma;
      double A[64] = {0}, g[8] = {0};
      for (int k = 0; k < ns; ++k) {
        if (!(un[k] < c)) continue;
        const int i = sup[k];
        const double t = 1.0 - (un[k] / c) * (un[k] / c), wt = t * t, iw = 1.0 / w[i], px = rx[i] + u[i], py = ry[i] + v[i];
        const double Jx[8] = {x[i] * iw, y[i] * iw, iw, 0, 0, 0, -px * x[i] * iw, -px * y[i] * iw};
        const double Jy[8] = {0, 0, 0, x[i] * iw, y[i] * iw, iw, -py * x[i] * iw, -py * y[i] * iw};
        for (int a = 0; a < npar; ++a) {
          g[a] += wt * (Jx[a] * rx[i] + Jy[a] * ry[i]);
          for (int b = 0; b < npar; ++b) A[a * npar + b] += wt * (Jx[a] * Jx[b] + Jy[a] * Jy[b]);
        }
      }
      if (!solve_n(npar, A, g)) break;
      double step = 0;
      for (int a = 0; a < npar; ++a) { h[a] -= g[a]; if (fabs(g[a]) > step) step = fabs(g[a]); }
      if (step < 1e-14) break;
    }
    free(srt);
    RES()
    int cnt = 0;
    const double t2 = (thr * sc) * (thr * sc);
    for (int i = 0; i < n; ++i) if (rx[i] * rx[i] + ry[i] * ry[i] <= t2) ++cnt;
    if (cnt >= min_pts) {
      mat3_mul(Ti, h, tmp); mat3_mul(tmp, T, Hout);
      const double inv = 1.0 / Hout[8];
      for (int i = 0; i < 9; ++i) Hout[i] *= inv;
      *n_inl = cnt;
      ok = 1;
    }
  }
#undef RES
  free(x); free(sup);
  return ok;
}

/* stabilo_ref.py:360-408. pq / pt: [n][2] float32 full-resolution pixels (query -> train). Returns 1 and H (row major) or 0. */
int stab_ransac(const float* pq, const float* pt, int n, int fw, int fh, float thr, int n_hyp, uint32_t seed, int affine, double* Hout, int* n_inl) {
  *n_inl = 0;
  const int ns = affine ? 3 : 4;
  if (n < ns) return 0;
  double* p = malloc(sizeof(double) * n * 4);
  double* q = p + 2 * n;
  for (int i = 0; i < 2 * n; ++i) { p[i] = pq[i]; q[i] = pt[i]; }
  const double cx = fw / 2.0, cy = fh / 2.0, sc = 2.0 / fw;
  const double thr2 = (double)(thr * thr);
  double T[9], Ti[9];
  norm_mats(cx, cy, sc, T, Ti);
  int64_t* cost = malloc(sizeof(int64_t) * n_hyp);
  double* Hs = malloc(sizeof(double) * 9 * n_hyp);
#pragma omp parallel for schedule(dynamic, 16)
  for (int hyp = 0; hyp < n_hyp; ++hyp) {
    cost[hyp] = INT64_MAX;
    int idx[4] = {-1, -1, -1, -1}, got = 0;
    uint32_t ctr = 0;
    while (got < ns) {
      const int c = (int)(hash_u32(seed ^ hash_u32((uint32_t)hyp * 977u + ctr)) % (uint32_t)n);
      ++ctr;
      int dup = 0;
      for (int j = 0; j < got; ++j) dup |= idx[j] == c;
      if (!dup) idx[got++] = c;
    }
    double x[4], y[4], u[4], v[4], hv[9];
    for (int k = 0; k < ns; ++k) { x[k] = (p[2 * idx[k]] - cx) * sc; y[k] = (p[2 * idx[k] + 1] - cy) * sc; u[k] = (q[2 * idx[k]] - cx) * sc; v[k] = (q[2 * idx[k] + 1] - cy) * sc; }
    if (affine) {
      double M1[9], M2[9], r1[3], r2[3];
      for (int k = 0; k < 3; ++k) { M1[3 * k] = M2[3 * k] = x[k]; M1[3 * k + 1] = M2[3 * k + 1] = y[k]; M1[3 * k + 2] = M2[3 * k + 2] = 1.0; r1[k] = u[k]; r2[k] = v[k]; }
      const double det = x[0] * (y[1] - y[2]) - y[0] * (x[1] - x[2]) + (x[1] * y[2] - x[2] * y[1]);
      if (!(fabs(det) > 1e-9)) continue;
      if (!solve_n(3, M1, r1) || !solve_n(3, M2, r2)) continue;
      hv[0] = r1[0]; hv[1] = r1[1]; hv[2] = r1[2]; hv[3] = r2[0]; hv[4] = r2[1]; hv[5] = r2[2]; hv[6] = 0; hv[7] = 0;
    } else {
      double A[64] = {0}, rhs[8];
      for (int i = 0; i < 4; ++i) {
        double* r0 = A + 16 * i; double* r1 = r0 + 8;
        r0[0] = x[i]; r0[1] = y[i]; r0[2] = 1; r0[6] = -u[i] * x[i]; r0[7] = -u[i] * y[i];
        r1[3] = x[i]; r1[4] = y[i]; r1[5] = 1; r1[6] = -v[i] * x[i]; r1[7] = -v[i] * y[i];
        rhs[2 * i] = u[i]; rhs[2 * i + 1] = v[i];
      }
      if (!solve_n(8, A, rhs)) continue;
      memcpy(hv, rhs, sizeof rhs);
    }
    hv[8] = 1.0;
    double tmp[9], H[9];
    mat3_mul(Ti, hv, tmp); mat3_mul(tmp, T, H);
    if (!(fabs(H[8]) > 1e-12)) continue;
    int64_t acc = 0;
    for (int i = 0; i < n; ++i) {
      const double X = p[2 * i], Y = p[2 * i + 1], w = H[6] * X + H[7] * Y + H[8];
      double e = thr2;
      if (fabs(w) > 1e-12) {
        const double dx = (H[0] * X + H[1] * Y + H[2]) / w - q[2 * i], dy = (H[3] * X + H[4] * Y + H[5]) / w - q[2 * i + 1];
        e = fmin(dx * dx + dy * dy, thr2);
      }
      acc += (int64_t)floor(e * 1024.0 + 0.5);
    }
    cost[hyp] = acc;
    memcpy(Hs + 9 * hyp, H, sizeof H);
  }
  int best = -1;
  for (int hyp = 0; hyp < n_hyp; ++hyp) if (cost[hyp] != INT64_MAX && (best < 0 || cost[hyp] < cost[best])) best = hyp;
  int ok = 0;
  if (best >= 0) {
    double H0[9];
    for (int i = 0; i < 9; ++i) H0[i] = Hs[9 * best + i] / Hs[9 * best + 8];
    ok = refine(H0, p, q, n, cx, cy, sc, (double)thr, affine, Hout, n_inl);
  }
  free(p); free(cost); free(Hs);
  return ok;
}
