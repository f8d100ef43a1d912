// ByteTrack / BoT-SORT association on the host (C++), the sequential part of the hot path.
//
// Stands in for ultralytics.trackers.{byte_tracker.BYTETracker, bot_sort.BOTSORT}.update and
// the lapx solver it calls (reference call site geotrax/extract.py:153 with persist=True;
// parameters geotrax/cfg/default.yaml:361-389). The algorithm is restated step for step in
// oracle/bytetrack_ref.py; both follow the published ByteTrack procedure as ultralytics ships
// it: Kalman predict -> (GMC) -> IoU(+score fusion) cost -> LAP with cost limit -> second
// association on low-score detections -> unconfirmed tracks -> new tracks -> lost/removed
// bookkeeping -> duplicate removal.
//
// The LAP: lap.lapjv(cost, extend_cost=True, cost_limit=t) minimises the assignment cost where
// leaving a row or a column unmatched costs t/2 each, i.e. a pair is only worth matching when
// cost < t. Pairs at or above the limit are dropped up front and the remaining sparse problem is
// solved exactly by successive shortest augmenting paths (see linear_assignment below): same
// optimum as the dense (n+m)^2 extended problem, a tiny fraction of the work.
#include "tracker.hpp"
#include "kalman.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <functional>
#include <unordered_set>
#include <vector>

namespace gtx {

namespace {
// -DGTX_TRK_PROF: per-phase wall time of update(), printed when the library unloads (tuning only)
#ifdef GTX_TRK_PROF
struct ProfAcc {
  double t[16] = {0}; const char* n[16] = {nullptr}; int calls = 0;
  double cnt[4] = {0};
  ~ProfAcc() { if (calls) { for (int i = 0; i < 16; ++i) if (n[i]) fprintf(stderr, "trk %-10s %8.1f us/call\n", n[i], 1e6 * t[i] / calls);
    fprintf(stderr, "trk sizes: pool %.1f dets_hi %.1f pairs %.1f lost %.1f\n", cnt[0] / calls, cnt[1] / calls, cnt[2] / calls, cnt[3] / calls); } }
};
static ProfAcc g_prof;
#define PROF_T0 auto prof_t = std::chrono::steady_clock::now(); int prof_i = 0; g_prof.calls++;
#define PROF_MARK(name) { auto now__ = std::chrono::steady_clock::now(); g_prof.t[prof_i] += std::chrono::duration<double>(now__ - prof_t).count(); g_prof.n[prof_i++] = #name; prof_t = now__; }
#else
#define PROF_T0
#define PROF_MARK(name)
#endif

enum State { kNew = 0, kTracked = 1, kLost = 2, kRemoved = 3 };

using trk::Kalman;

struct Track {
  float tlwh0[4];       // detection box (top-left, w, h), float32 like STrack._tlwh
  double mean[8];
  double cov[64];
  bool has_mean = false;
  bool activated = false;
  int state = kNew;
  float score = 0;
  int cls = 0;
  int idx = 0;          // index of the detection inside the frame's detection list
  int id = 0;
  int frame_id = 0, start_frame = 0, tracklet_len = 0;
  // FastTracker (type 4) only: the filter's recent means (newest last), occlusion state
  std::vector<std::array<double, 8>> hist;
  bool occluded = false, occ_lost = false;
  int occ_frames = 0;
  // BoT-SORT with_reid only (BOTrack): the detection's normalised appearance vector and the track's 0.9-EMA of them (float32)
  std::vector<float> curr_feat, smooth_feat;
  void update_features(const std::vector<float>& f) {   // BOTrack.update_features (feat arrives normalised: it is a curr_feat)
    curr_feat = f;
    if (smooth_feat.empty()) { smooth_feat = f; }
    else for (size_t i = 0; i < f.size(); ++i) smooth_feat[i] = 0.9f * smooth_feat[i] + (1.f - 0.9f) * f[i];
    float ss = 0.f;
    for (float v : smooth_feat) ss += v * v;
    const float nrm = std::sqrt(ss);
    for (float& v : smooth_feat) v /= nrm;
  }
};

// ---- exact sparse LAP ----
// lap.lapjv(cost, extend_cost=True, cost_limit=L) semantics: minimise the sum of matched costs
// where leaving a row or a column unmatched costs L/2 each. Up to a constant that is
//     minimise  sum over matched (c_ij - L),   every row either matched or "skipped" at cost 0,
// so only pairs with c_ij < L can ever be matched. Solved exactly by successive shortest
// augmenting paths (Dijkstra on reduced costs with row/column potentials) over the sparse list of
// feasible pairs; every row owns a private zero-cost skip column, so a search never leaves the
// row's connected component. On traffic scenes components are a handful of boxes; the cost is
// O(E log E) in the number of feasible pairs even when hundreds of boxes overlap.
// cost: rows x cols, row-major float32 (as numpy hands it to lapjv). x[r] = matched col or -1,
// y[c] = matched row or -1.
struct SparseCost {                    // feasible pairs (cost < limit) of a rows x cols problem, CSR by row, columns ascending
  int rows = 0, cols = 0;
  std::vector<int> start, adj;
  std::vector<double> w;             // cost - limit (< 0)
};

void linear_assignment_sparse(const SparseCost& P, std::vector<int>& x, std::vector<int>& y);

void linear_assignment(const std::vector<float>& cost, int rows, int cols, double limit, std::vector<int>& x,
                       std::vector<int>& y) {
  SparseCost P;
  P.rows = rows; P.cols = cols;
  P.start.assign(rows + 1, 0);
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      const double v = (double)cost[(size_t)r * cols + c];
      if (v < limit) { P.adj.push_back(c); P.w.push_back(v - limit); }
    }
    P.start[r + 1] = (int)P.adj.size();
  }
  linear_assignment_sparse(P, x, y);
}

void linear_assignment_sparse(const SparseCost& P, std::vector<int>& x, std::vector<int>& y) {
  const int rows = P.rows, cols = P.cols;
  const std::vector<int>&start = P.start, &adj = P.adj;
  const std::vector<double>& w = P.w;
  x.assign(rows, -1);
  y.assign(cols, -1);
  if (rows == 0 || cols == 0) return;
  const int ncol = cols + rows;                       // real columns, then one skip column per row
  std::vector<double> u(rows, 0.0), v(ncol, 0.0), dist(ncol);
  std::vector<int> col_match(ncol, -1), row_match(rows, -1), parent(ncol), seen_list;
  std::vector<char> done(ncol, 0), touched(ncol, 0);
  for (int r = 0; r < rows; ++r) {                    // feasible start: reduced costs >= 0
    double m = 0.0;
    for (int e = start[r]; e < start[r + 1]; ++e) m = std::min(m, w[e]);
    u[r] = m;
  }
  using Item = std::pair<double, int>;
  std::vector<Item> heap;
  auto push = [&](double d, int j) { heap.emplace_back(d, j); std::push_heap(heap.begin(), heap.end(), std::greater<Item>()); };
  for (int i = 0; i < rows; ++i) {
    if (start[i] == start[i + 1]) continue;           // no feasible pair: stays unmatched
    {
      // Fast path: the first column a Dijkstra search from row i would finalise is the row's cheapest reduced
      // cost (ties: lowest column, as the heap orders pairs); when that column is still free the search ends there
      // and only u[i] changes. Most rows of a traffic scene end this way; the general search below is unchanged.
      double best = 0.0 - u[i] - v[cols + i];         // the row's own skip column
      int bj = cols + i;
      for (int e = start[i]; e < start[i + 1]; ++e) {
        const int j = adj[e];
        const double nd = w[e] - u[i] - v[j];
        if (nd < best || (nd == best && j < bj)) { best = nd; bj = j; }
      }
      if (col_match[bj] < 0) {
        u[i] += best;
        col_match[bj] = i;
        row_match[i] = bj;
        continue;
      }
    }
    heap.clear();
    seen_list.clear();
    auto relax_row = [&](int r, double base, int via_col) {
      for (int e = start[r]; e < start[r + 1]; ++e) {
        const int j = adj[e];
        if (done[j]) continue;
        const double nd = base + (w[e] - u[r] - v[j]);
        if (!touched[j] || nd < dist[j]) {
          if (!touched[j]) { touched[j] = 1; seen_list.push_back(j); }
          dist[j] = nd; parent[j] = via_col; push(nd, j);
        }
      }
      const int sj = cols + r;                         // the row's own skip column, weight 0
      if (!done[sj]) {
        const double nd = base + (0.0 - u[r] - v[sj]);
        if (!touched[sj] || nd < dist[sj]) {
          if (!touched[sj]) { touched[sj] = 1; seen_list.push_back(sj); }
          dist[sj] = nd; parent[sj] = via_col; push(nd, sj);
        }
      }
    };
    relax_row(i, 0.0, -1);
    int jf = -1;
    double D = 0.0;
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), std::greater<Item>());
      const Item it = heap.back();
      heap.pop_back();
      const int j = it.second;
      if (done[j] || it.first > dist[j]) continue;
      done[j] = 1;
      if (col_match[j] < 0) { jf = j; D = dist[j]; break; }
      relax_row(col_match[j], dist[j], j);
    }
    // potentials: finalised columns and the rows matched to them (plus the start row)
    u[i] += D;
    for (int j : seen_list) {
      if (done[j] && j != jf) {
        const double delta = D - dist[j];
        v[j] -= delta;
        u[col_match[j]] += delta;
      }
    }
    // augment along the parent chain
    for (int j = jf; j >= 0;) {
      const int pj = parent[j];
      const int r = pj < 0 ? i : col_match[pj];
      col_match[j] = r;
      row_match[r] = j;
      j = pj;
    }
    for (int j : seen_list) { done[j] = 0; touched[j] = 0; }
  }
  for (int r = 0; r < rows; ++r)
    if (row_match[r] >= 0 && row_match[r] < cols) { x[r] = row_match[r]; y[row_match[r]] = r; }
}

inline void xyxy_of(const Track& t, bool xywh_state, float out[4]) {
  double tl[4];
  if (!t.has_mean) {
    for (int i = 0; i < 4; ++i) tl[i] = t.tlwh0[i];
  } else if (xywh_state) {
    tl[2] = t.mean[2]; tl[3] = t.mean[3];
    tl[0] = t.mean[0] - tl[2] / 2; tl[1] = t.mean[1] - tl[3] / 2;
  } else {
    tl[3] = t.mean[3]; tl[2] = t.mean[2] * t.mean[3];
    tl[0] = t.mean[0] - tl[2] / 2; tl[1] = t.mean[1] - tl[3] / 2;
  }
  if (!t.has_mean) {
    // float32 arithmetic on the float32 _tlwh, like numpy does
    out[0] = t.tlwh0[0]; out[1] = t.tlwh0[1];
    out[2] = t.tlwh0[2] + t.tlwh0[0]; out[3] = t.tlwh0[3] + t.tlwh0[1];
  } else {
    out[0] = (float)tl[0]; out[1] = (float)tl[1]; out[2] = (float)(tl[2] + tl[0]); out[3] = (float)(tl[3] + tl[1]);
  }
}

}  // namespace

void lap_limited(const float* cost, int rows, int cols, double limit, std::vector<int>& x, std::vector<int>& y) {
  linear_assignment(std::vector<float>(cost, cost + (size_t)rows * cols), rows, cols, limit, x, y);
}

// Dense rectangular assignment, rows <= cols after an optional transpose: the classic shortest-augmenting-path Hungarian
// method with row / column potentials (O(rows^2 cols), array scans only -- the sparse solver above pays a heap and an
// edge list per feasible pair, which for a dense 130 x 140 OC-SORT cost matrix was 0.7 ms per frame).
void lap_full(const std::vector<double>& cost, int rows, int cols, std::vector<int>& x) {
  x.assign(rows, -1);
  if (rows == 0 || cols == 0) return;
  const bool tr = rows > cols;
  const int n = tr ? cols : rows, m = tr ? rows : cols;              // n <= m
  auto c = [&](int i, int j) { return tr ? cost[(size_t)j * cols + i] : cost[(size_t)i * cols + j]; };
  const double inf = 1e300;
  std::vector<double> u(n + 1, 0.0), v(m + 1, 0.0), minv(m + 1);
  std::vector<int> p(m + 1, 0), way(m + 1, 0);
  std::vector<char> used(m + 1);
  for (int i = 1; i <= n; ++i) {
    p[0] = i;
    int j0 = 0;
    std::fill(minv.begin(), minv.end(), inf);
    std::fill(used.begin(), used.end(), 0);
    do {
      used[j0] = 1;
      const int i0 = p[j0];
      double delta = inf;
      int j1 = 0;
      for (int j = 1; j <= m; ++j) {
        if (used[j]) continue;
        const double cur = c(i0 - 1, j - 1) - u[i0] - v[j];
        if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
        if (minv[j] < delta) { delta = minv[j]; j1 = j; }
      }
      for (int j = 0; j <= m; ++j) {
        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
        else minv[j] -= delta;
      }
      j0 = j1;
    } while (p[j0] != 0);
    do {
      const int j1 = way[j0];
      p[j0] = p[j1];
      j0 = j1;
    } while (j0);
  }
  for (int j = 1; j <= m; ++j) {
    if (p[j] == 0) continue;
    if (tr) x[j - 1] = p[j] - 1;      // transposed: "rows" of the solver are the caller's columns
    else x[p[j] - 1] = j - 1;
  }
}

struct ByteTracker::Impl {
  gtx_tracker_config cfg;
  Kalman kf;
  std::vector<Track> tracked, lost;
  std::vector<char> id_seen;          // scratch byte map over track ids
  // ultralytics' removed_stracks is only ever read for its ids (sub_stracks): keep the ids, in order, plus a set
  std::vector<int> removed_order;
  std::unordered_set<int> removed_set;
  int frame_id = 0;
  int next_id = 0;
  int max_time_lost = 30;

  int new_id() { return ++next_id; }

  void measurement(const float tlwh[4], double z[4]) const {
    // tlwh_to_xyah / tlwh_to_xywh on float32 input (numpy float32 arithmetic)
    float r[4] = {tlwh[0], tlwh[1], tlwh[2], tlwh[3]};
    r[0] += r[2] / 2;
    r[1] += r[3] / 2;
    if (!kf.xywh) r[2] /= r[3];
    for (int i = 0; i < 4; ++i) z[i] = r[i];
  }
  void tlwh_of(const Track& t, float out[4]) const {
    if (!t.has_mean) { std::memcpy(out, t.tlwh0, sizeof(float) * 4); return; }
    double w, h;
    if (kf.xywh) { w = t.mean[2]; h = t.mean[3]; } else { h = t.mean[3]; w = t.mean[2] * t.mean[3]; }
    out[0] = (float)(t.mean[0] - w / 2); out[1] = (float)(t.mean[1] - h / 2); out[2] = (float)w; out[3] = (float)h;
  }

  // 1 - IoU (float32 like matching.iou_distance), optionally fused with detection scores
  std::vector<float> dists(const std::vector<Track*>& a, const std::vector<Track*>& b, bool fuse) const {
    std::vector<float> d(a.size() * b.size());
    std::vector<float> ab(a.size() * 4), bb(b.size() * 4);
    for (size_t i = 0; i < a.size(); ++i) xyxy_of(*a[i], kf.xywh, &ab[i * 4]);
    for (size_t j = 0; j < b.size(); ++j) xyxy_of(*b[j], kf.xywh, &bb[j * 4]);
    const size_t nb = b.size();
    std::vector<float> barea(nb), bscore(nb);
    for (size_t j = 0; j < nb; ++j) {
      const float* q = &bb[j * 4];
      barea[j] = (q[2] - q[0]) * (q[3] - q[1]);
      bscore[j] = b[j]->score;
    }
    for (size_t i = 0; i < a.size(); ++i) {
      const float* p = &ab[i * 4];
      const float a1 = (p[2] - p[0]) * (p[3] - p[1]);
      float* row = &d[i * nb];
      for (size_t j = 0; j < nb; ++j) {
        const float* q = &bb[j * 4];
        // disjoint boxes (the vast majority): iw or ih clamps to 0, so iou = 0 / den = 0 and the cost
        // (fused or not) is exactly 1 for every positive finite den -- skip the arithmetic
        const float den0 = barea[j] + a1 + 1e-7f;
        // bitwise | on purpose: one well-predicted branch instead of four data-dependent ones
        const bool skip = ((q[0] >= p[2]) | (q[2] <= p[0]) | (q[1] >= p[3]) | (q[3] <= p[1])) &
                          (den0 > 0.f) & (den0 < std::numeric_limits<float>::infinity());
        if (skip) { row[j] = 1.f; continue; }
        const float iw = std::max(0.f, std::min(p[2], q[2]) - std::max(p[0], q[0]));
        const float ih = std::max(0.f, std::min(p[3], q[3]) - std::max(p[1], q[1]));
        const float inter = iw * ih;
        const float iou = inter / (barea[j] + a1 - inter + 1e-7f);
        float cost = 1.f - iou;
        if (fuse) {
          const float sim = (1.f - cost) * bscore[j];
          cost = 1.f - sim;
        }
        row[j] = cost;
      }
    }
    return d;
  }

  // Feasible pairs of the IoU(+score) cost, i.e. the entries of dists(a, b, fuse) below `limit`, without
  // forming the dense matrix: a pair of disjoint boxes costs exactly 1 (see dists) and every limit used is
  // below 1, so only boxes that overlap can qualify. Boxes are swept in x order; the cost of an
  // overlapping pair is the same float32 expression as in dists().
  mutable std::vector<int> sc_ob;
  mutable std::vector<float> sc_x1, sc_y1, sc_x2, sc_y2, sc_area, sc_score, sc_cost;
  mutable std::vector<std::pair<int, float>> sc_rowbuf;
  SparseCost sparse_costs(const std::vector<Track*>& a, const std::vector<Track*>& b, bool fuse, double limit) const {
    SparseCost P;
    const int na = (int)a.size(), nb = (int)b.size();
    P.rows = na; P.cols = nb;
    P.start.assign(na + 1, 0);
    if (na == 0 || nb == 0) return P;
    std::vector<float> ab((size_t)na * 4), bb((size_t)nb * 4);
    for (int i = 0; i < na; ++i) xyxy_of(*a[i], kf.xywh, &ab[(size_t)i * 4]);
    for (int j = 0; j < nb; ++j) xyxy_of(*b[j], kf.xywh, &bb[(size_t)j * 4]);
    bool regular = true;           // the sweep needs finite, non-inverted boxes; anything else takes the dense path
    for (float v : ab) regular &= std::isfinite(v);
    for (float v : bb) regular &= std::isfinite(v);
    for (int i = 0; i < na && regular; ++i) regular &= ab[i * 4 + 2] >= ab[i * 4] && ab[i * 4 + 3] >= ab[i * 4 + 1];
    for (int j = 0; j < nb && regular; ++j) regular &= bb[j * 4 + 2] >= bb[j * 4] && bb[j * 4 + 3] >= bb[j * 4 + 1];
    if (!regular || limit > 1.0) {
      const std::vector<float> d = dists(a, b, fuse);
      for (int r = 0; r < na; ++r) {
        for (int c = 0; c < nb; ++c) {
          const double v = (double)d[(size_t)r * nb + c];
          if (v < limit) { P.adj.push_back(c); P.w.push_back(v - limit); }
        }
        P.start[r + 1] = (int)P.adj.size();
      }
      return P;
    }
    // b sorted by x1, structure of arrays: the candidates of a row are a contiguous band of the sorted list
    // (x1_a - widest_b < x1_b < x2_a), and the cost of the whole band is computed without branches so that the
    // compiler vectorises it (same float32 operations in the same order as dists(); a pair that does not
    // overlap comes out at cost 1 and is dropped by the limit like before).
    std::vector<int>& ob = sc_ob;
    ob.resize(nb);
    for (int j = 0; j < nb; ++j) ob[j] = j;
    std::sort(ob.begin(), ob.end(), [&](int p, int q) { return bb[p * 4] < bb[q * 4]; });
    sc_x1.resize(nb); sc_y1.resize(nb); sc_x2.resize(nb); sc_y2.resize(nb); sc_area.resize(nb); sc_score.resize(nb);
    float bw_max = 0.f;                       // widest box of b: x2_b > x1_a implies x1_b > x1_a - bw_max
    for (int k = 0; k < nb; ++k) {
      const float* q = &bb[(size_t)ob[k] * 4];
      sc_x1[k] = q[0]; sc_y1[k] = q[1]; sc_x2[k] = q[2]; sc_y2[k] = q[3];
      sc_area[k] = (q[2] - q[0]) * (q[3] - q[1]);
      sc_score[k] = b[ob[k]]->score;
      bw_max = std::max(bw_max, q[2] - q[0]);
    }
    sc_cost.resize(nb);
    const float* __restrict__ X1 = sc_x1.data(); const float* __restrict__ Y1 = sc_y1.data();
    const float* __restrict__ X2 = sc_x2.data(); const float* __restrict__ Y2 = sc_y2.data();
    const float* __restrict__ AR = sc_area.data(); const float* __restrict__ SC = sc_score.data();
    float* __restrict__ CO = sc_cost.data();
    std::vector<std::pair<int, float>>& rowbuf = sc_rowbuf;
    P.adj.reserve((size_t)na * 6); P.w.reserve((size_t)na * 6);
    for (int i = 0; i < na; ++i) {
      const float* p = &ab[(size_t)i * 4];
      const float p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
      const float a1 = (p2 - p0) * (p3 - p1);
      const int hi = (int)(std::lower_bound(sc_x1.begin(), sc_x1.end(), p2) - sc_x1.begin());
      const int lo = (int)(std::lower_bound(sc_x1.begin(), sc_x1.begin() + hi, p0 - bw_max) - sc_x1.begin());
      if (fuse) {
#pragma clang loop vectorize(enable) interleave(enable)
        for (int k = lo; k < hi; ++k) {
          const float iw = std::max(0.f, std::min(p2, X2[k]) - std::max(p0, X1[k]));
          const float ih = std::max(0.f, std::min(p3, Y2[k]) - std::max(p1, Y1[k]));
          const float inter = iw * ih;
          const float iou = inter / (AR[k] + a1 - inter + 1e-7f);
          const float cost = 1.f - iou;
          const float sim = (1.f - cost) * SC[k];
          CO[k] = 1.f - sim;
        }
      } else {
#pragma clang loop vectorize(enable) interleave(enable)
        for (int k = lo; k < hi; ++k) {
          const float iw = std::max(0.f, std::min(p2, X2[k]) - std::max(p0, X1[k]));
          const float ih = std::max(0.f, std::min(p3, Y2[k]) - std::max(p1, Y1[k]));
          const float inter = iw * ih;
          const float iou = inter / (AR[k] + a1 - inter + 1e-7f);
          CO[k] = 1.f - iou;
        }
      }
      rowbuf.clear();
      for (int k = lo; k < hi; ++k)
        if ((double)CO[k] < limit) rowbuf.emplace_back(ob[k], CO[k]);
      std::sort(rowbuf.begin(), rowbuf.end());
      for (const auto& e : rowbuf) { P.adj.push_back(e.first); P.w.push_back((double)e.second - limit); }
      P.start[i + 1] = (int)P.adj.size();
    }
    return P;
  }

  // BOTSORT.get_dists with the appearance branch: IoU cost, its proximity mask, score fusion, then min with cosine distance / 2
  // where the latter is accepted (<= 1 - appearance_thresh) and the boxes are proximate. Dense: the branch is opt-in.
  SparseCost reid_costs(const std::vector<Track*>& a, const std::vector<Track*>& b, bool fuse, double limit) const {
    SparseCost P;
    const int na = (int)a.size(), nb = (int)b.size();
    P.rows = na; P.cols = nb;
    P.start.assign(na + 1, 0);
    if (na == 0 || nb == 0) return P;
    const std::vector<float> d0 = dists(a, b, false), df = fuse ? dists(a, b, true) : d0;
    const float far = 1.f - cfg.proximity_thresh;
    const double reject = 1.0 - (double)cfg.appearance_thresh;
    for (int r = 0; r < na; ++r) {
      for (int c = 0; c < nb; ++c) {
        double v = (double)df[(size_t)r * nb + c];
        const std::vector<float>&u = a[r]->smooth_feat, &w = b[c]->curr_feat;
        if (!(d0[(size_t)r * nb + c] > far) && !u.empty() && u.size() == w.size()) {
          double uw = 0, uu = 0, ww = 0;                       // scipy cdist(..., 'cosine') on float32 rows, in double
          for (size_t k = 0; k < u.size(); ++k) { uw += (double)u[k] * w[k]; uu += (double)u[k] * u[k]; ww += (double)w[k] * w[k]; }
          double e = std::max(0.0, 1.0 - uw / (std::sqrt(uu) * std::sqrt(ww))) / 2.0;
          if (e > reject) e = 1.0;
          v = std::min(v, e);
        }
        if (v < limit) { P.adj.push_back(c); P.w.push_back(v - limit); }
      }
      P.start[r + 1] = (int)P.adj.size();
    }
    return P;
  }

  void activate(Track& t) {
    t.id = new_id();
    double z[4];
    measurement(t.tlwh0, z);
    kf.initiate(z, t.mean, t.cov);
    t.has_mean = true;
    t.tracklet_len = 0;
    t.state = kTracked;
    if (frame_id == 1) t.activated = true;
    t.frame_id = frame_id;
    t.start_frame = frame_id;
  }
  void absorb(Track& t, const Track& det, bool re_activate) {
    float tl[4];
    tlwh_of(det, tl);
    double z[4];
    measurement(tl, z);
    kf.update(t.mean, t.cov, z);
    if (re_activate) t.tracklet_len = 0; else t.tracklet_len += 1;
    t.state = kTracked;
    t.activated = true;
    t.frame_id = frame_id;
    t.score = det.score;
    t.cls = det.cls;
    t.idx = det.idx;
    t.occluded = false; t.occ_lost = false; t.occ_frames = 0;
    if (!det.curr_feat.empty()) t.update_features(det.curr_feat);   // BOTrack.update / re_activate
  }
  // FastTracker: the largest fraction of t's box that one box of `others` covers
  double covered(const Track& t, const std::vector<Track*>& others) const {
    float a[4];
    xyxy_of(t, kf.xywh, a);
    const double area = std::max((double)(a[2] - a[0]) * (double)(a[3] - a[1]), 1e-12);
    double best = 0.0;
    for (const Track* o : others) {
      if (o == &t) continue;
      float b[4];
      xyxy_of(*o, kf.xywh, b);
      const double iw = (double)std::min(a[2], b[2]) - (double)std::max(a[0], b[0]), ih = (double)std::min(a[3], b[3]) - (double)std::max(a[1], b[1]);
      if (iw > 0 && ih > 0) best = std::max(best, iw * ih / area);
    }
    return best;
  }
};

ByteTracker::ByteTracker(const gtx_tracker_config& cfg) : impl_(new Impl) {
  GTX_CHECK(cfg.type == 0 || cfg.type == 1 || cfg.type == 4, "tracker type %d: bytetrack (0), botsort (1) and fasttrack (4) live here", cfg.type);
  impl_->cfg = cfg;
  impl_->kf.xywh = cfg.type == 1;
  const int fr = cfg.frame_rate > 0 ? cfg.frame_rate : 30;
  impl_->max_time_lost = (int)(fr / 30.0 * cfg.track_buffer);
}
ByteTracker::~ByteTracker() = default;

void ByteTracker::reset() {
  impl_->tracked.clear();
  impl_->lost.clear();
  impl_->removed_order.clear();
  impl_->removed_set.clear();
  impl_->frame_id = 0;
  impl_->next_id = 0;
}

void ByteTracker::update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap,
                         int* n_out, float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx,
                         const float* feats, int feat_dim) {
  Impl& S = *impl_;
  const gtx_tracker_config& A = S.cfg;
  const bool reid = A.type == 1 && A.with_reid != 0;
  GTX_CHECK(!reid || n == 0 || (feats != nullptr && feat_dim > 0), "tracker: with_reid needs an appearance vector per detection (gtx_tracker_update_feats)");
  S.frame_id += 1;
  PROF_T0

  // detections -> candidate tracks (xyxy2xywh then xywh2ltwh in float32)
  std::vector<Track> det_hi, det_lo;
  for (int i = 0; i < n; ++i) {
    const float* b = xyxy + 4 * i;
    const float cx = (b[0] + b[2]) / 2, cy = (b[1] + b[3]) / 2, w = b[2] - b[0], h = b[3] - b[1];
    Track t;
    t.tlwh0[0] = cx - w / 2; t.tlwh0[1] = cy - h / 2; t.tlwh0[2] = w; t.tlwh0[3] = h;
    t.score = conf[i];
    t.cls = cls[i];
    t.idx = i;
    if (reid) {                                     // BOTrack.__init__ -> update_features: feat /= ||feat|| in float32
      t.curr_feat.assign(feats + (size_t)i * feat_dim, feats + (size_t)(i + 1) * feat_dim);
      float ss = 0.f;
      for (float v : t.curr_feat) ss += v * v;
      const float nrm = std::sqrt(ss);
      for (float& v : t.curr_feat) v /= nrm;
      t.smooth_feat = t.curr_feat;
    }
    if (conf[i] >= A.track_high_thresh) det_hi.push_back(t);
    else if (conf[i] > A.track_low_thresh) det_lo.push_back(t);
  }

  PROF_MARK(dets)
  std::vector<Track*> unconfirmed, confirmed;
  for (Track& t : S.tracked) (t.activated ? confirmed : unconfirmed).push_back(&t);
  // pool = joint(confirmed, lost)
  std::vector<Track*> pool = confirmed;
  {                                   // joint_stracks: first occurrence of an id wins (ids are small integers: a byte map)
    std::vector<char>& seen = S.id_seen;
    if (seen.size() < (size_t)S.next_id + 2) seen.resize((size_t)S.next_id + 1026, 0);   // all zero between uses
    for (Track* q : pool) seen[q->id] = 1;
    for (Track& t : S.lost)
      if (!seen[t.id]) { seen[t.id] = 1; pool.push_back(&t); }
    for (Track* q : pool) seen[q->id] = 0;
  }
  PROF_MARK(pool)
  // Kalman predict (velocity of the size/aspect state is zeroed for non-tracked tracks)
  for (Track* t : pool) {
    if (t->state != kTracked) {
      if (S.kf.xywh) { t->mean[6] = 0; t->mean[7] = 0; } else { t->mean[7] = 0; }
    }
    S.kf.predict(t->mean, t->cov);
  }
  if (A.type == 1 && gmc) {
    // STrack.multi_gmc: mean <- kron(I4, R) mean (+t on xy), cov <- R8 cov R8^T
    const double R[4] = {gmc[0], gmc[1], gmc[3], gmc[4]}, tx = gmc[2], ty = gmc[5];
    auto apply = [&](Track* t) {
      double m[8];
      for (int b = 0; b < 4; ++b) {
        m[2 * b] = R[0] * t->mean[2 * b] + R[1] * t->mean[2 * b + 1];
        m[2 * b + 1] = R[2] * t->mean[2 * b] + R[3] * t->mean[2 * b + 1];
      }
      m[0] += tx; m[1] += ty;
      std::memcpy(t->mean, m, sizeof m);
      double tmp[64], out[64];
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
          const int bi = i / 2, ri = i % 2;
          tmp[i * 8 + j] = R[ri * 2] * t->cov[(2 * bi) * 8 + j] + R[ri * 2 + 1] * t->cov[(2 * bi + 1) * 8 + j];
        }
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
          const int bj = j / 2, rj = j % 2;
          out[i * 8 + j] = tmp[i * 8 + 2 * bj] * R[rj * 2] + tmp[i * 8 + 2 * bj + 1] * R[rj * 2 + 1];
        }
      std::memcpy(t->cov, out, sizeof out);
    };
    for (Track* t : pool) apply(t);
    for (Track* t : unconfirmed) apply(t);
  }

  PROF_MARK(predict)
  std::vector<Track> activated_new;             // tracks created this frame
  std::vector<Track*> activated, refind, lost_now, removed_now, ft_unmatched;
  std::vector<int> x, y;

  // ---- first association: pool vs high-score detections ----
  std::vector<Track*> dh;
  for (Track& d : det_hi) dh.push_back(&d);
  {
    const SparseCost c1 = reid ? S.reid_costs(pool, dh, A.fuse_score != 0, A.match_thresh) : S.sparse_costs(pool, dh, A.fuse_score != 0, A.match_thresh);
#ifdef GTX_TRK_PROF
    g_prof.cnt[0] += pool.size(); g_prof.cnt[1] += dh.size(); g_prof.cnt[2] += c1.adj.size(); g_prof.cnt[3] += S.lost.size();
#endif
    PROF_MARK(cost1)
    linear_assignment_sparse(c1, x, y);
  }
  PROF_MARK(assoc1)
  std::vector<int> u_track, u_det;
  for (size_t i = 0; i < pool.size(); ++i) {
    if (x[i] >= 0) {
      Track* t = pool[i];
      if (t->state == kTracked) { S.absorb(*t, *dh[x[i]], false); activated.push_back(t); }
      else { S.absorb(*t, *dh[x[i]], true); refind.push_back(t); }
    } else u_track.push_back((int)i);
  }
  for (size_t j = 0; j < dh.size(); ++j)
    if (y[j] < 0) u_det.push_back((int)j);

  PROF_MARK(absorb1)
  // ---- second association: remaining tracked tracks vs low-score detections (plain IoU, 0.5) ----
  std::vector<Track*> r_tracked, dl;
  for (int i : u_track)
    if (pool[i]->state == kTracked) r_tracked.push_back(pool[i]);
  for (Track& d : det_lo) dl.push_back(&d);
  linear_assignment_sparse(S.sparse_costs(r_tracked, dl, false, 0.5), x, y);
  for (size_t i = 0; i < r_tracked.size(); ++i) {
    Track* t = r_tracked[i];
    if (x[i] >= 0) {
      if (t->state == kTracked) { S.absorb(*t, *dl[x[i]], false); activated.push_back(t); }
      else { S.absorb(*t, *dl[x[i]], true); refind.push_back(t); }
    } else if (A.type == 4) {
      ft_unmatched.push_back(t);                 // FastTracker: occluded or lost, decided below against the tracks that did find a detection
    } else if (t->state != kLost) {
      t->state = kLost;
      lost_now.push_back(t);
    }
  }
  if (A.type == 4) {
    // tracker.fasttrack (default.yaml:426-443; oracle/fasttrack_ref.py states the choices made where the description leaves them open): a
    // confirmed track without a detection whose box another active track covers by occ_cover_thresh stays active on its prediction for up
    // to active_occ_to_lost_thresh frames; at the onset its filter is rolled back (velocity of reset_velocity_offset_occ frames ago, position
    // of reset_pos_offset_occ frames ago carried forward), the velocity dampened and the box enlarged once.
    std::vector<Track*> seen = activated;
    seen.insert(seen.end(), refind.begin(), refind.end());
    const int vel_off = std::max(A.reset_velocity_offset_occ, 0), pos_off = std::max(A.reset_pos_offset_occ, 0);
    for (Track* t : ft_unmatched) {
      if (S.covered(*t, seen) >= (double)A.occ_cover_thresh && t->occ_frames < A.active_occ_to_lost_thresh) {
        if (!t->occluded) {
          t->occluded = true;
          if (!t->hist.empty()) {
            const int nh = (int)t->hist.size();
            const double* pv = vel_off > 0 ? t->hist[std::max(nh - vel_off, 0)].data() : t->mean;
            double v[4] = {pv[4], pv[5], pv[6], pv[7]};
            const int k = std::min(pos_off, nh);
            if (k > 0) {
              const double* pp = t->hist[nh - k].data();
              for (int i = 0; i < 4; ++i) t->mean[i] = pp[i] + (k + 1) * v[i];   // k stored frames back + this frame's prediction step
            }
            for (int i = 0; i < 4; ++i) t->mean[4 + i] = v[i];
          }
          for (int i = 0; i < 4; ++i) t->mean[4 + i] *= (double)A.dampen_motion_occ;
          t->mean[3] *= (double)A.enlarge_bbox_occ;
        }
        t->occ_frames += 1;
        t->idx = -1;
        activated.push_back(t);
      } else {
        if (t->occluded) t->occ_lost = true;
        t->occluded = false;
        if (t->state != kLost) { t->state = kLost; lost_now.push_back(t); }
      }
    }
  }

  PROF_MARK(second)
  // ---- unconfirmed tracks vs leftover high-score detections (0.7) ----
  std::vector<Track*> dleft;
  for (int j : u_det) dleft.push_back(dh[j]);
  linear_assignment_sparse(reid ? S.reid_costs(unconfirmed, dleft, A.fuse_score != 0, 0.7) : S.sparse_costs(unconfirmed, dleft, A.fuse_score != 0, 0.7), x, y);
  for (size_t i = 0; i < unconfirmed.size(); ++i) {
    if (x[i] >= 0) { S.absorb(*unconfirmed[i], *dleft[x[i]], false); activated.push_back(unconfirmed[i]); }
    else { unconfirmed[i]->state = kRemoved; removed_now.push_back(unconfirmed[i]); }
  }
  // ---- new tracks ----
  for (size_t j = 0; j < dleft.size(); ++j) {
    if (y[j] >= 0) continue;
    Track t = *dleft[j];
    if (t.score < A.new_track_thresh) continue;
    if (A.type == 4 && A.init_iou_suppress < 1.f) {     // no new track on top of an active one (tracks that found a detection this frame)
      bool on_top = false;
      float a[4] = {t.tlwh0[0], t.tlwh0[1], t.tlwh0[0] + t.tlwh0[2], t.tlwh0[1] + t.tlwh0[3]};
      for (Track* o : activated) {
        if (o->idx < 0) continue;                       // occluded tracks run on their prediction: they do not suppress
        float b[4];
        xyxy_of(*o, S.kf.xywh, b);
        const float iw = std::max(std::min(a[2], b[2]) - std::max(a[0], b[0]), 0.f), ih = std::max(std::min(a[3], b[3]) - std::max(a[1], b[1]), 0.f);
        const float inter = iw * ih, uni = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
        if (1.f - (1.f - inter / (uni + 1e-7f)) >= A.init_iou_suppress) { on_top = true; break; }
      }
      if (!on_top)
        for (Track* o : refind) {
          float b[4];
          xyxy_of(*o, S.kf.xywh, b);
          const float iw = std::max(std::min(a[2], b[2]) - std::max(a[0], b[0]), 0.f), ih = std::max(std::min(a[3], b[3]) - std::max(a[1], b[1]), 0.f);
          const float inter = iw * ih, uni = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
          if (1.f - (1.f - inter / (uni + 1e-7f)) >= A.init_iou_suppress) { on_top = true; break; }
        }
      if (on_top) continue;
    }
    S.activate(t);
    activated_new.push_back(t);
  }
  // ---- lost tracks that timed out ----
  for (Track& t : S.lost)
    if (S.frame_id - t.frame_id > ((A.type == 4 && t.occ_lost) ? A.occ_reappear_window : S.max_time_lost)) { t.state = kRemoved; removed_now.push_back(&t); }

  PROF_MARK(third)
  // ---- state update (order of the lists is part of the output contract) ----
  auto has_ptr = [](const std::vector<Track*>& v, const Track* p) { return std::find(v.begin(), v.end(), p) != v.end(); };
  std::vector<Track> new_tracked, new_lost;
  new_tracked.reserve(S.tracked.size() + 16);
  struct IdMap {                       // ids present in new_tracked
    std::vector<char>& m;
    bool count(int id) const { return m[id] != 0; }
    void insert(int id) { m[id] = 1; }
  } ids{S.id_seen};
  if (S.id_seen.size() < (size_t)S.next_id + 2) S.id_seen.resize((size_t)S.next_id + 1026, 0);   // activate() may have handed out new ids
  for (Track& t : S.tracked)
    if (t.state == kTracked) { new_tracked.push_back(t); ids.insert(t.id); }
  // joint(tracked, activated): `activated` holds existing tracks (already in tracked if they were
  // tracked before) and the new ones, in the order they were appended upstream
  // -> first-association matches, second-association matches, unconfirmed matches, new tracks.
  for (Track* t : activated)
    if (!ids.count(t->id)) { ids.insert(t->id); new_tracked.push_back(*t); }
  for (Track& t : activated_new)
    if (!ids.count(t.id)) { ids.insert(t.id); new_tracked.push_back(t); }
  for (Track* t : refind)
    if (!ids.count(t->id)) { ids.insert(t->id); new_tracked.push_back(*t); }
  // lost = sub(lost, tracked) + lost_now, then minus removed
  // ultralytics subtracts self.removed_stracks *before* extending it with this frame's removals
  const std::unordered_set<int>& removed_ids = S.removed_set;
  new_lost.reserve(S.lost.size() + lost_now.size());
  for (Track& t : S.lost)
    if (!ids.count(t.id) && !has_ptr(lost_now, &t) && !removed_ids.count(t.id)) new_lost.push_back(t);
  for (Track* t : lost_now)
    if (!removed_ids.count(t->id)) new_lost.push_back(*t);
  for (const Track& t : new_tracked) S.id_seen[t.id] = 0;     // the byte map is all zero between uses
  PROF_MARK(lists)
  // duplicates between tracked and lost (IoU distance < 0.15): keep the older track
  {
    std::vector<Track*> pa, pb;
    for (Track& t : new_tracked) pa.push_back(&t);
    for (Track& t : new_lost) pb.push_back(&t);
    const SparseCost pd = S.sparse_costs(pa, pb, false, (double)0.15f);   // pairs with 1 - IoU < 0.15
    std::vector<char> dupa(pa.size(), 0), dupb(pb.size(), 0);
    for (size_t p = 0; p < pa.size(); ++p)
      for (int e = pd.start[p]; e < pd.start[p + 1]; ++e) {
        const int q = pd.adj[e];
        const int tp = pa[p]->frame_id - pa[p]->start_frame, tq = pb[q]->frame_id - pb[q]->start_frame;
        if (tp > tq) dupb[q] = 1; else dupa[p] = 1;
      }
    if (!pd.adj.empty()) {               // usual frame: no duplicates, nothing to rebuild
      std::vector<Track> ta, tb;
      ta.reserve(pa.size()); tb.reserve(pb.size());
      for (size_t p = 0; p < pa.size(); ++p) if (!dupa[p]) ta.push_back(*pa[p]);
      for (size_t q = 0; q < pb.size(); ++q) if (!dupb[q]) tb.push_back(*pb[q]);
      new_tracked.swap(ta);
      new_lost.swap(tb);
    }
  }
  // lost tracks that were marked removed this frame stay in `lost` until the next frame's
  // subtraction (upstream behaviour); keep their state so they never match again.
  PROF_MARK(dups)
  S.tracked.swap(new_tracked);
  S.lost.swap(new_lost);
  for (Track* t : removed_now)
    if (S.removed_set.insert(t->id).second) S.removed_order.push_back(t->id);
  if (S.removed_order.size() > 1000) {                      // upstream keeps the 999 most recent
    const size_t drop = S.removed_order.size() - 999;
    for (size_t i = 0; i < drop; ++i) S.removed_set.erase(S.removed_order[i]);
    S.removed_order.erase(S.removed_order.begin(), S.removed_order.begin() + drop);
  }

  if (A.type == 4) {
    const size_t keep = (size_t)std::max(std::max(A.reset_velocity_offset_occ, A.reset_pos_offset_occ), 0) + 1;
    for (Track& t : S.tracked) {
      std::array<double, 8> m;
      std::memcpy(m.data(), t.mean, sizeof(double) * 8);
      t.hist.push_back(m);
      if (t.hist.size() > keep) t.hist.erase(t.hist.begin());
    }
  }
  int k = 0;
  for (const Track& t : S.tracked) {
    if (!t.activated) continue;
    if (k < cap) {
      float b[4];
      xyxy_of(t, S.kf.xywh, b);
      if (out_xyxy) std::memcpy(out_xyxy + 4 * k, b, sizeof b);
      if (out_id) out_id[k] = t.id;
      if (out_score) out_score[k] = t.score;
      if (out_cls) out_cls[k] = t.cls;
      if (out_det_idx) out_det_idx[k] = t.idx;
    }
    ++k;
  }
  PROF_MARK(out)
  *n_out = std::min(k, cap);
}

}  // namespace gtx
