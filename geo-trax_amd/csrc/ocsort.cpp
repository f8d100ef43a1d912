// OC-SORT (Cao et al., "Observation-Centric SORT", CVPR 2023) behind the gtx_tracker C ABI (type 2): the tracker the
// reference selects with `tracker.active: ocsort` (geotrax/cfg/default.yaml:391-404; handed to ultralytics by
// geotrax/utils/config_utils.py:127-194 and run inside model.track(), geotrax/extract.py:153).
// Host C++, float64 throughout; statement by statement the procedure of oracle/ocsort_ref.py (which restates the
// authors' public implementation): per-track 7-state constant-velocity Kalman filter with the observation-centric
// re-update (ORU: freeze at the first missed frame, re-run along the straight virtual trajectory when the track is
// observed again), velocity-direction consistency in the first association (OCM), optional BYTE pass, and a second
// association of the leftovers against the tracks' last observations (OCR). Parameter mapping as in the oracle's header:
// det_thresh = track_high_thresh, iou_threshold = 1 - match_thresh, max_age = track_buffer, min_hits = 3.
//
// Type 3 (`tracker.active: deepocsort`, default.yaml:406-427) is the same tracker with Deep OC-SORT's camera-motion
// compensation (Maggiolino et al., "Deep OC-SORT", ICIP 2023): before the prediction step the frame's 2x3 warp (the GMC
// of gmc.hip, `gmc_method: sparseOptFlow`) moves every track into the new frame's coordinates -- the Kalman position and
// velocity with their covariance blocks, the frozen state of a lost track, the last observation, and the observations
// the velocity direction is taken from. The appearance branch (with_reid) is not built; the host wrapper refuses it.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "tracker.hpp"
#include "vec_acos.hpp"

namespace gtx {

namespace {

struct Box5 { double v[5]; };   // x1, y1, x2, y2, score

inline void to_z(const double* b, double z[4]) {
  const double w = b[2] - b[0], h = b[3] - b[1];
  z[0] = b[0] + w / 2.0; z[1] = b[1] + h / 2.0; z[2] = w * h; z[3] = w / (h + 1e-6);
}
inline void to_box(const double* x, double b[4]) {
  const double w = std::sqrt(x[2] * x[3]), h = x[2] / w;
  b[0] = x[0] - w / 2.0; b[1] = x[1] - h / 2.0; b[2] = x[0] + w / 2.0; b[3] = x[1] + h / 2.0;
}
inline double iou_of(const double* a, const double* b) {
  const double xx1 = std::max(a[0], b[0]), yy1 = std::max(a[1], b[1]), xx2 = std::min(a[2], b[2]), yy2 = std::min(a[3], b[3]);
  const double wh = std::max(0.0, xx2 - xx1) * std::max(0.0, yy2 - yy1);
  return wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
}

// filterpy's Kalman filter as the authors extend it (KalmanFilterNew): constant matrices are written out.
struct OcKalman {
  double x[7];
  double P[49];
  bool observed = false, has_saved = false;
  double sx[7], sP[49];
  int s_hist = 0;            // history length when frozen (includes the missed frame's entry)
  int n_hist = 0;            // observations + misses recorded so far
  int seen_idx = -1;         // history index and value of the last observation
  double seen_z[4] = {0, 0, 0, 0};

  OcKalman() {
    std::memset(x, 0, sizeof x);
    std::memset(P, 0, sizeof P);
    for (int i = 0; i < 7; ++i) P[i * 7 + i] = i < 4 ? 10.0 : 10000.0;
  }
  static double q(int i) { return i < 4 ? 1.0 : (i < 6 ? 0.01 : 0.0001); }
  static double r(int i) { return i < 2 ? 1.0 : 10.0; }

  void predict() {
    // x = F x, P = F P F^T + Q with F = I + e_0 e_4^T + e_1 e_5^T + e_2 e_6^T
    double Fx[7];
    for (int i = 0; i < 7; ++i) Fx[i] = x[i] + (i < 3 ? x[i + 4] : 0.0);
    std::memcpy(x, Fx, sizeof Fx);
    double FP[49];
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) FP[i * 7 + j] = P[i * 7 + j] + (i < 3 ? P[(i + 4) * 7 + j] : 0.0);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) P[i * 7 + j] = FP[i * 7 + j] + (j < 3 ? FP[i * 7 + j + 4] : 0.0) + (i == j ? q(i) : 0.0);
  }

  // Deep OC-SORT's apply_affine_correction for the (u, v, s, r) filter: position by m and t, velocity by m, their
  // covariance blocks by m . m^T; the same for the frozen prior of a lost track and for the centre of the observation ORU
  // will start its virtual trajectory from. Area and aspect ratio are left alone, as in the authors' code.
  void apply_affine(const double m[4], const double t[2]) {
    affine_state(x, P, m, t);
    if (!observed && has_saved) affine_state(sx, sP, m, t);
    if (seen_idx >= 0) {
      const double u = seen_z[0], v = seen_z[1];
      seen_z[0] = m[0] * u + m[1] * v + t[0];
      seen_z[1] = m[2] * u + m[3] * v + t[1];
    }
  }

  void update_none() {
    ++n_hist;
    if (observed) {            // first frame without an observation: freeze the prior of this frame
      std::memcpy(sx, x, sizeof x);
      std::memcpy(sP, P, sizeof P);
      s_hist = n_hist;
      has_saved = true;
    }
    observed = false;
  }

  void update(const double z[4]) {
    const int idx = n_hist++;
    if (!observed && has_saved) {
      unfreeze(idx, z);
    } else {
      seen_idx = idx;
      std::memcpy(seen_z, z, sizeof seen_z);
    }
    observed = true;
    // y = z - Hx, S = H P H^T + R, K = P H^T S^-1, x += K y, P = (I - KH) P (I - KH)^T + K R K^T
    double y[4], S[16], Si[16], K[28];
    for (int i = 0; i < 4; ++i) y[i] = z[i] - x[i];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) S[i * 4 + j] = P[i * 7 + j] + (i == j ? r(i) : 0.0);
    invert4(S, Si);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += P[i * 7 + k] * Si[k * 4 + j];
        K[i * 4 + j] = s;
      }
    for (int i = 0; i < 7; ++i) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += K[i * 4 + k] * y[k];
      x[i] += s;
    }
    double A[49], AP[49], Pn[49];            // A = I - K H
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) A[i * 7 + j] = (i == j ? 1.0 : 0.0) - (j < 4 ? K[i * 4 + j] : 0.0);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) {
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s += A[i * 7 + k] * P[k * 7 + j];
        AP[i * 7 + j] = s;
      }
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) {
        double s = 0.0;
        for (int k = 0; k < 7; ++k) s += AP[i * 7 + k] * A[j * 7 + k];
        double krk = 0.0;
        for (int k = 0; k < 4; ++k) krk += K[i * 4 + k] * r(k) * K[j * 4 + k];
        Pn[i * 7 + j] = s + krk;
      }
    std::memcpy(P, Pn, sizeof Pn);
  }

 private:
  static void affine_state(double* xs, double* Ps, const double m[4], const double t[2]) {
    const double u = xs[0], v = xs[1], du = xs[4], dv = xs[5];
    xs[0] = m[0] * u + m[1] * v + t[0]; xs[1] = m[2] * u + m[3] * v + t[1];
    xs[4] = m[0] * du + m[1] * dv;      xs[5] = m[2] * du + m[3] * dv;
    for (int o = 0; o <= 4; o += 4) {                 // the 2x2 blocks at (0,0) and (4,4): B <- m B m^T
      const double b00 = Ps[o * 7 + o], b01 = Ps[o * 7 + o + 1], b10 = Ps[(o + 1) * 7 + o], b11 = Ps[(o + 1) * 7 + o + 1];
      const double c00 = m[0] * b00 + m[1] * b10, c01 = m[0] * b01 + m[1] * b11, c10 = m[2] * b00 + m[3] * b10, c11 = m[2] * b01 + m[3] * b11;
      Ps[o * 7 + o] = c00 * m[0] + c01 * m[1];       Ps[o * 7 + o + 1] = c00 * m[2] + c01 * m[3];
      Ps[(o + 1) * 7 + o] = c10 * m[0] + c11 * m[1]; Ps[(o + 1) * 7 + o + 1] = c10 * m[2] + c11 * m[3];
    }
  }

  // ORU: back to the frozen state, then predict/update along the straight line between the last observation before the
  // gap and the new one (box centre, width and height interpolated); the caller's regular update with z follows.
  void unfreeze(int idx2, const double z2[4]) {
    const int idx1 = seen_idx;
    double z1[4];
    std::memcpy(z1, seen_z, sizeof z1);
    std::memcpy(x, sx, sizeof x);
    std::memcpy(P, sP, sizeof P);
    n_hist = s_hist - 1;
    observed = true;
    const double w1 = std::sqrt(z1[2] * z1[3]), h1 = std::sqrt(z1[2] / z1[3]);
    const double w2 = std::sqrt(z2[2] * z2[3]), h2 = std::sqrt(z2[2] / z2[3]);
    const int gap = idx2 - idx1;
    const double dx = (z2[0] - z1[0]) / gap, dy = (z2[1] - z1[1]) / gap, dw = (w2 - w1) / gap, dh = (h2 - h1) / gap;
    for (int i = 0; i < gap; ++i) {
      const double w = w1 + (i + 1) * dw, h = h1 + (i + 1) * dh;
      const double zv[4] = {z1[0] + (i + 1) * dx, z1[1] + (i + 1) * dy, w * h, w / h};
      update(zv);
      if (i != gap - 1) predict();
    }
  }

  static void invert4(const double* M, double* inv) {   // Gauss-Jordan with partial pivoting
    double a[4][8];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) { a[i][j] = M[i * 4 + j]; a[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
      int p = c;
      for (int r2 = c + 1; r2 < 4; ++r2)
        if (std::fabs(a[r2][c]) > std::fabs(a[p][c])) p = r2;
      if (p != c)
        for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]);
      const double d = 1.0 / a[c][c];
      for (int j = 0; j < 8; ++j) a[c][j] *= d;
      for (int r2 = 0; r2 < 4; ++r2) {
        if (r2 == c) continue;
        const double f = a[r2][c];
        if (f != 0.0)
          for (int j = 0; j < 8; ++j) a[r2][j] -= f * a[c][j];
      }
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
  }
};

struct OcTrack {
  OcKalman kf;
  int id = 0;
  int tsu = 0, hits = 0, streak = 0, age = 0;
  Box5 last_obs{{-1, -1, -1, -1, -1}};
  std::vector<std::pair<int, Box5>> obs;    // (age, observation), ascending age; only the last delta_t + 1 ages matter
  bool has_vel = false;
  double vel[2] = {0, 0};                    // (dy, dx), unit length
  float score = 0.f;
  int cls = 0, idx = -1;
  std::vector<double> emb;                   // Deep OC-SORT with_reid: unit-length appearance vector, dynamic-alpha EMA of the matched detections'
  void update_emb(const std::vector<double>& e, double alpha) {
    if (emb.empty()) { emb = e; return; }
    double ss = 0.0;
    for (size_t i = 0; i < e.size(); ++i) { emb[i] = alpha * emb[i] + (1.0 - alpha) * e[i]; ss += emb[i] * emb[i]; }
    const double nrm = std::sqrt(ss);
    for (double& v : emb) v /= nrm;
  }

  bool seen() const { return last_obs.v[0] + last_obs.v[1] + last_obs.v[2] + last_obs.v[3] + last_obs.v[4] >= 0; }
  const Box5* obs_at(int a) const {
    for (auto it = obs.rbegin(); it != obs.rend(); ++it) {
      if (it->first == a) return &it->second;
      if (it->first < a) break;
    }
    return nullptr;
  }
};

// c[d] = clamp(vx * dx + vy * dy, -1, 1) with (dx, dy) the unit vector from (cx2, cy2) to detection d's centre (norm + 1e-6, as
// the authors write it); the scalar expression of the loop it replaces, operation by operation.
inline void direction_cos_block(const double* cx, const double* cy, int n4, double cx2, double cy2, double vx, double vy, double* c) {
  const __m256d x2 = _mm256_set1_pd(cx2), y2 = _mm256_set1_pd(cy2), vxv = _mm256_set1_pd(vx), vyv = _mm256_set1_pd(vy);
  const __m256d eps = _mm256_set1_pd(1e-6), one = _mm256_set1_pd(1.0), mone = _mm256_set1_pd(-1.0);
  for (int i = 0; i < n4; i += 4) {
    __m256d dx = _mm256_sub_pd(_mm256_loadu_pd(cx + i), x2), dy = _mm256_sub_pd(_mm256_loadu_pd(cy + i), y2);
    const __m256d nrm = _mm256_add_pd(_mm256_sqrt_pd(_mm256_add_pd(_mm256_mul_pd(dx, dx), _mm256_mul_pd(dy, dy))), eps);
    dx = _mm256_div_pd(dx, nrm);
    dy = _mm256_div_pd(dy, nrm);
    const __m256d dot = _mm256_add_pd(_mm256_mul_pd(vxv, dx), _mm256_mul_pd(vyv, dy));
    _mm256_storeu_pd(c + i, _mm256_min_pd(one, _mm256_max_pd(mone, dot)));
  }
}

inline void direction(const double* b1, const double* b2, double out[2]) {
  const double cx1 = (b1[0] + b1[2]) / 2.0, cy1 = (b1[1] + b1[3]) / 2.0, cx2 = (b2[0] + b2[2]) / 2.0, cy2 = (b2[1] + b2[3]) / 2.0;
  const double dy = cy2 - cy1, dx = cx2 - cx1, n = std::sqrt(dy * dy + dx * dx) + 1e-6;
  out[0] = dy / n; out[1] = dx / n;
}

struct Det { Box5 b; int cls, idx; std::vector<double> emb; double alpha = 0.0; };

}  // namespace

struct OcSortTracker::Impl {
  gtx_tracker_config cfg;
  double det_thresh, low, new_thr, iou_thr, inertia;
  int max_age, delta_t, min_hits;
  bool use_byte, cmc = false;
  bool reid = false;                         // type 3 with gtx_tracker_config.with_reid
  double prox = 0.5, app_thr = 0.9, alpha_fixed = 0.95;
  static constexpr double kWEmb = 0.75, kAwBottom = 0.5;   // Deep OC-SORT's w_association_emb and aw_param (no config key: the authors' defaults)
  std::vector<OcTrack> trackers;
  int frame_count = 0, next_id = 0;

  void track_update(OcTrack& t, const Det* d) {
    if (!d) { t.kf.update_none(); return; }
    if (t.seen()) {
      const Box5* prev = nullptr;
      for (int i = 0; i < delta_t && !prev; ++i) prev = t.obs_at(t.age - (delta_t - i));
      if (!prev) prev = &t.last_obs;
      direction(prev->v, d->b.v, t.vel);
      t.has_vel = true;
    }
    t.last_obs = d->b;
    t.obs.emplace_back(t.age, d->b);
    if ((int)t.obs.size() > delta_t + 2) t.obs.erase(t.obs.begin(), t.obs.end() - (delta_t + 2));
    t.tsu = 0;
    ++t.hits;
    ++t.streak;
    t.score = (float)d->b.v[4]; t.cls = d->cls; t.idx = d->idx;
    double z[4];
    to_z(d->b.v, z);
    t.kf.update(z);
  }
  void track_update_emb(OcTrack& t, const Det& d) const {
    if (reid && !d.emb.empty()) t.update_emb(d.emb, d.alpha);
  }

  // camera-motion compensation of one track (type 3): g = [m00 m01 t0; m10 m11 t1], previous frame -> this frame
  void track_affine(OcTrack& t, const double* g) const {
    const double m[4] = {g[0], g[1], g[3], g[4]}, tv[2] = {g[2], g[5]};
    auto corners = [&](Box5& b) {
      const double x1 = b.v[0], y1 = b.v[1], x2 = b.v[2], y2 = b.v[3];
      b.v[0] = m[0] * x1 + m[1] * y1 + tv[0]; b.v[1] = m[2] * x1 + m[3] * y1 + tv[1];
      b.v[2] = m[0] * x2 + m[1] * y2 + tv[0]; b.v[3] = m[2] * x2 + m[3] * y2 + tv[1];
    };
    if (t.seen()) corners(t.last_obs);
    for (auto& o : t.obs)
      if (o.first >= t.age - delta_t) corners(o.second);     // the ages the velocity direction can still be taken from
    t.kf.apply_affine(m, tv);
  }

  void track_predict(OcTrack& t, double box[4]) {
    if (t.kf.x[6] + t.kf.x[2] <= 0) t.kf.x[6] *= 0.0;
    t.kf.predict();
    ++t.age;
    if (t.tsu > 0) t.streak = 0;
    ++t.tsu;
    to_box(t.kf.x, box);
  }

  Box5 k_previous(const OcTrack& t) const {
    if (t.obs.empty()) return Box5{{-1, -1, -1, -1, -1}};
    for (int i = 0; i < delta_t; ++i)
      if (const Box5* p = t.obs_at(t.age - (delta_t - i))) return *p;
    return t.obs.back().second;
  }

  // First association (OCM). dets: high-score detections; trks: predicted boxes.
  void associate(const std::vector<Det>& dets, const std::vector<std::array<double, 4>>& trks, std::vector<std::pair<int, int>>& matches,
                 std::vector<int>& u_d, std::vector<int>& u_t) const {
    const int n = (int)dets.size(), m = (int)trks.size();
    matches.clear(); u_d.clear(); u_t.clear();
    if (m == 0) { for (int d = 0; d < n; ++d) u_d.push_back(d); return; }
    if (n == 0) { for (int t = 0; t < m; ++t) u_t.push_back(t); return; }
    std::vector<double> iou((size_t)n * m);
    std::vector<int> row_cnt(n, 0), col_cnt(m, 0);
    {
      // iou_of() over all pairs, one detection against every track per sweep (tracks as five arrays: the sweep vectorises)
      std::vector<double> tx1(m), ty1(m), tx2(m), ty2(m), ta(m);
      for (int t = 0; t < m; ++t) {
        const double* b = trks[t].data();
        tx1[t] = b[0]; ty1[t] = b[1]; tx2[t] = b[2]; ty2[t] = b[3];
        ta[t] = (b[2] - b[0]) * (b[3] - b[1]);
      }
      for (int d = 0; d < n; ++d) {
        const double* a = dets[d].b.v;
        const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], aa = (a[2] - a[0]) * (a[3] - a[1]);
        double* row = iou.data() + (size_t)d * m;
        for (int t = 0; t < m; ++t) {
          const double xx1 = std::max(a0, tx1[t]), yy1 = std::max(a1, ty1[t]), xx2 = std::min(a2, tx2[t]), yy2 = std::min(a3, ty2[t]);
          const double wh = std::max(0.0, xx2 - xx1) * std::max(0.0, yy2 - yy1);
          row[t] = wh / (aa + ta[t] - wh);
        }
        for (int t = 0; t < m; ++t)
          if (row[t] > iou_thr) { ++row_cnt[d]; ++col_cnt[t]; }
      }
    }
    std::vector<int> x(n, -1);
    const int rmax = *std::max_element(row_cnt.begin(), row_cnt.end()), cmax = *std::max_element(col_cnt.begin(), col_cnt.end());
    if (rmax == 1 && cmax == 1) {              // every candidate pair is unambiguous: no assignment problem to solve
      for (int d = 0; d < n; ++d)
        for (int t = 0; t < m; ++t)
          if (iou[(size_t)d * m + t] > iou_thr) x[d] = t;
    } else {
      // the velocity-direction term of every (detection, track) pair: only needed when there is an assignment to solve
      // One track against every detection at a time, four detections per AVX2 lane group: cosine of the angle between the
      // track's direction and the direction to the detection, then acos over the row (vec_acos.hpp; libm's acos alone was
      // 0.3 ms of a 0.8 ms frame at 130 x 140 pairs).
      std::vector<double> cost((size_t)n * m);
      const int n4 = (n + 3) & ~3;
      std::vector<double> dcx(n4, 0.0), dcy(n4, 0.0), cosv(n4), acv(n4);
      for (int d = 0; d < n; ++d) {
        const double* b = dets[d].b.v;
        dcx[d] = (b[0] + b[2]) / 2.0; dcy[d] = (b[1] + b[3]) / 2.0;
      }
      for (int t = 0; t < m; ++t) {
        const OcTrack& tr = trackers[t];
        const Box5 po = k_previous(tr);
        if (!(po.v[4] >= 0)) {                       // no earlier observation: the term is multiplied by zero
          for (int d = 0; d < n; ++d) cost[(size_t)d * m + t] = -(iou[(size_t)d * m + t] + 0.0);
          continue;
        }
        const double cx2 = (po.v[0] + po.v[2]) / 2.0, cy2 = (po.v[1] + po.v[3]) / 2.0;
        const double vy = tr.has_vel ? tr.vel[0] : 0.0, vx = tr.has_vel ? tr.vel[1] : 0.0;
        direction_cos_block(dcx.data(), dcy.data(), n4, cx2, cy2, vx, vy, cosv.data());
        acos_block(cosv.data(), acv.data(), n4);
        for (int d = 0; d < n; ++d) {
          const double ang = (M_PI / 2.0 - std::fabs(acv[d])) / M_PI;
          cost[(size_t)d * m + t] = -(iou[(size_t)d * m + t] + ang * inertia * dets[d].b.v[4]);
        }
      }
      if (reid) {
        // Deep OC-SORT's appearance term: cosine similarity of the unit vectors, kept only for pairs the config's two gates let
        // through (IoU >= proximity_thresh -- the authors' own gate is IoU > 0 --, similarity >= appearance_thresh), weighted
        // adaptively (compute_aw_max_metric: a row / column whose two best similarities are close counts for less)
        std::vector<double> emb((size_t)n * m, 0.0), wgt((size_t)n * m, kWEmb);
        for (int d = 0; d < n; ++d)
          for (int t = 0; t < m; ++t) {
            const std::vector<double>&u = dets[d].emb, &w = trackers[t].emb;
            if (u.empty() || w.empty() || u.size() != w.size()) continue;
            double dot = 0.0;
            for (size_t k = 0; k < u.size(); ++k) dot += u[k] * w[k];
            const double io = iou[(size_t)d * m + t];
            if (io <= 0.0 || io < prox || dot < app_thr) continue;
            emb[(size_t)d * m + t] = dot;
          }
        auto two_best = [](const std::vector<double>& v, size_t off, size_t step, int cnt, double& b0, double& b1) {
          b0 = b1 = -1e300;
          for (int i = 0; i < cnt; ++i) {
            const double e = v[off + (size_t)i * step];
            if (e > b0) { b1 = b0; b0 = e; } else if (e > b1) { b1 = e; }
          }
        };
        if (m >= 2)
          for (int d = 0; d < n; ++d) {
            double b0, b1;
            two_best(emb, (size_t)d * m, 1, m, b0, b1);
            const double rw = b0 == 0.0 ? 0.0 : 1.0 - std::max(b1 / b0 - kAwBottom, 0.0) / (1.0 - kAwBottom);
            for (int t = 0; t < m; ++t) wgt[(size_t)d * m + t] *= rw;
          }
        if (n >= 2)
          for (int t = 0; t < m; ++t) {
            double b0, b1;
            two_best(emb, (size_t)t, (size_t)m, n, b0, b1);
            const double cw = b0 == 0.0 ? 0.0 : 1.0 - std::max(b1 / b0 - kAwBottom, 0.0) / (1.0 - kAwBottom);
            for (int d = 0; d < n; ++d) wgt[(size_t)d * m + t] *= cw;
          }
        for (size_t i = 0; i < cost.size(); ++i) cost[i] -= wgt[i] * emb[i];
      }
      lap_full(cost, n, m, x);
    }
    std::vector<char> t_used(m, 0);
    std::vector<int> back_d, back_t;
    for (int d = 0; d < n; ++d) {
      if (x[d] < 0) { u_d.push_back(d); continue; }
      t_used[x[d]] = 1;
    }
    for (int t = 0; t < m; ++t)
      if (!t_used[t]) u_t.push_back(t);
    for (int d = 0; d < n; ++d) {
      if (x[d] < 0) continue;
      if (iou[(size_t)d * m + x[d]] < iou_thr) { u_d.push_back(d); u_t.push_back(x[d]); }
      else matches.emplace_back(d, x[d]);
    }
  }
};

OcSortTracker::OcSortTracker(const gtx_tracker_config& cfg) : impl_(new Impl) {
  Impl& S = *impl_;
  S.cfg = cfg;
  S.det_thresh = cfg.track_high_thresh;
  S.low = cfg.track_low_thresh;
  S.new_thr = cfg.new_track_thresh;
  S.iou_thr = 1.0 - (double)cfg.match_thresh;
  S.inertia = cfg.inertia;
  S.max_age = cfg.track_buffer;
  S.delta_t = std::max(cfg.delta_t, 1);
  S.min_hits = cfg.min_hits > 0 ? cfg.min_hits : 3;
  S.use_byte = cfg.use_byte != 0;
  S.cmc = cfg.type == 3;
  S.reid = cfg.type == 3 && cfg.with_reid != 0;
  S.prox = cfg.proximity_thresh;
  S.app_thr = cfg.appearance_thresh;
  S.alpha_fixed = cfg.alpha_fixed_emb > 0.f ? cfg.alpha_fixed_emb : 0.95;
}
OcSortTracker::~OcSortTracker() = default;

void OcSortTracker::reset() {
  impl_->trackers.clear();
  impl_->frame_count = 0;
  impl_->next_id = 0;
}

void OcSortTracker::update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap,
                           int* n_out, float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx, const float* feats, int feat_dim) {
  Impl& S = *impl_;
  GTX_CHECK(!S.reid || n == 0 || (feats != nullptr && feat_dim > 0), "deepocsort: with_reid needs an appearance vector per detection (gtx_tracker_update_feats)");
  ++S.frame_count;
  std::vector<Det> dets, second;
  for (int i = 0; i < n; ++i) {
    const double c = (double)conf[i];
    Det d{{{(double)xyxy[4 * i], (double)xyxy[4 * i + 1], (double)xyxy[4 * i + 2], (double)xyxy[4 * i + 3], c}}, cls[i], i, {}, 0.0};
    if (S.reid && c > S.det_thresh) {            // unit-length vector; dynamic appearance: a confident detection moves the track's vector more
      d.emb.resize(feat_dim);
      double ss = 0.0;
      for (int k = 0; k < feat_dim; ++k) { d.emb[k] = (double)feats[(size_t)i * feat_dim + k]; ss += d.emb[k] * d.emb[k]; }
      const double nrm = std::sqrt(ss);
      for (double& v : d.emb) v /= nrm;
      const double trust = (c - S.det_thresh) / (1.0 - S.det_thresh);
      d.alpha = S.alpha_fixed + (1.0 - S.alpha_fixed) * (1.0 - trust);
    }
    if (c > S.det_thresh) dets.push_back(d);
    else if (c > S.low && c < S.det_thresh) second.push_back(d);
  }
  if (S.cmc && gmc)
    for (OcTrack& t : S.trackers) S.track_affine(t, gmc);
  // predictions; a track whose predicted box is not a number is dropped
  std::vector<std::array<double, 4>> trks;
  {
    std::vector<OcTrack> keep;
    keep.reserve(S.trackers.size());
    for (OcTrack& t : S.trackers) {
      std::array<double, 4> b;
      S.track_predict(t, b.data());
      if (std::isnan(b[0]) || std::isnan(b[1]) || std::isnan(b[2]) || std::isnan(b[3])) continue;
      trks.push_back(b);
      keep.push_back(std::move(t));
    }
    S.trackers.swap(keep);
  }
  std::vector<Box5> last_boxes;
  last_boxes.reserve(S.trackers.size());
  for (const OcTrack& t : S.trackers) last_boxes.push_back(t.last_obs);

  std::vector<std::pair<int, int>> matches;
  std::vector<int> u_d, u_t;
  S.associate(dets, trks, matches, u_d, u_t);
  for (auto& m : matches) { S.track_update(S.trackers[m.second], &dets[m.first]); S.track_update_emb(S.trackers[m.second], dets[m.first]); }

  auto second_round = [&](const std::vector<Det>& cand, const std::vector<int>& cand_idx, bool against_last, std::vector<int>& used_c) {
    // IoU of the candidates with the leftover tracks (predicted boxes, or last observations for OCR), assignment on -IoU
    const int a = (int)cand_idx.size(), b = (int)u_t.size();
    std::vector<double> io((size_t)a * b), neg((size_t)a * b);
    double mx = 0.0;
    for (int i = 0; i < a; ++i)
      for (int j = 0; j < b; ++j) {
        const double* tb = against_last ? last_boxes[u_t[j]].v : trks[u_t[j]].data();
        const double v = iou_of(cand[cand_idx[i]].b.v, tb);
        io[(size_t)i * b + j] = v;
        neg[(size_t)i * b + j] = -v;
        mx = (i == 0 && j == 0) ? v : std::max(mx, v);
      }
    if (!(mx > S.iou_thr)) return;
    std::vector<int> x;
    lap_full(neg, a, b, x);
    std::vector<int> done_t;
    for (int i = 0; i < a; ++i) {
      if (x[i] < 0 || io[(size_t)i * b + x[i]] < S.iou_thr) continue;
      S.track_update(S.trackers[u_t[x[i]]], &cand[cand_idx[i]]);
      if (against_last) S.track_update_emb(S.trackers[u_t[x[i]]], cand[cand_idx[i]]);   // OCR updates the vector too, the BYTE pass does not
      done_t.push_back(u_t[x[i]]);
      used_c.push_back(cand_idx[i]);
    }
    std::vector<int> rest;
    for (int t : u_t)
      if (std::find(done_t.begin(), done_t.end(), t) == done_t.end()) rest.push_back(t);
    u_t.swap(rest);
  };
  if (S.use_byte && !second.empty() && !u_t.empty()) {
    std::vector<int> all(second.size()), used;
    for (size_t i = 0; i < second.size(); ++i) all[i] = (int)i;
    second_round(second, all, false, used);
  }
  if (!u_d.empty() && !u_t.empty()) {
    std::vector<int> used;
    second_round(dets, u_d, true, used);
    std::vector<int> rest;
    for (int d : u_d)
      if (std::find(used.begin(), used.end(), d) == used.end()) rest.push_back(d);
    u_d.swap(rest);
  }
  for (int t : u_t) S.track_update(S.trackers[t], nullptr);
  std::sort(u_d.begin(), u_d.end());
  for (int d : u_d) {
    if (dets[d].b.v[4] < S.new_thr) continue;
    OcTrack t;
    double z[4];
    to_z(dets[d].b.v, z);
    std::memcpy(t.kf.x, z, sizeof z);
    t.id = ++S.next_id;
    t.score = (float)dets[d].b.v[4]; t.cls = dets[d].cls; t.idx = dets[d].idx;
    if (S.reid) t.emb = dets[d].emb;
    S.trackers.push_back(std::move(t));
  }
  // output, newest track first; tracks not updated for more than max_age frames leave
  int k = 0;
  std::vector<char> drop(S.trackers.size(), 0);
  for (int i = (int)S.trackers.size() - 1; i >= 0; --i) {
    const OcTrack& t = S.trackers[i];
    if (t.tsu < 1 && (t.streak >= S.min_hits || S.frame_count <= S.min_hits) && k < cap) {
      double b[4];
      if (!t.seen()) to_box(t.kf.x, b);
      else std::memcpy(b, t.last_obs.v, sizeof b);
      for (int j = 0; j < 4; ++j) out_xyxy[4 * k + j] = (float)b[j];
      out_id[k] = t.id; out_score[k] = t.score; out_cls[k] = t.cls; out_det_idx[k] = t.idx;
      ++k;
    }
    if (t.tsu > S.max_age) drop[i] = 1;
  }
  *n_out = k;
  std::vector<OcTrack> keep;
  keep.reserve(S.trackers.size());
  for (size_t i = 0; i < S.trackers.size(); ++i)
    if (!drop[i]) keep.push_back(std::move(S.trackers[i]));
  S.trackers.swap(keep);
}

}  // namespace gtx
