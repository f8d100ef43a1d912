// TrackTrack (tracker.tracktrack of the reference's config, default.yaml:445-470) on the host: multi-cue cost (height-modulated IoU,
// confidence, corner angle) with iterative mutual-minimum assignment under a shrinking threshold and track-aware initialisation.
//
// The port the reference runs lives in ultralytics (not vendored, not installed here); this is written from the config's own
// description of every parameter and from the published method it names. oracle/tracktrack_ref.py restates the same procedure
// step for step and lists the CHOICES made where the description leaves one open (one pool of tracked + lost tracks against
// all detections above track_low_thresh, no "deleted" detections to penalise, KalmanFilterXYWH, ...). Same update() contract as
// the other trackers; the camera-motion warp is applied like BoT-SORT's.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "kalman.hpp"
#include "tracker.hpp"

namespace gtx {

namespace {
constexpr double kInf = std::numeric_limits<double>::infinity();
enum { kTracked = 1, kLost = 2, kRemoved = 3 };

struct Obs { int frame; double b[4]; };
struct TTrack {
  double det[4];                 // the detection's box (xyxy)
  double mean[8], cov[64];
  bool has_mean = false;
  int state = kTracked;
  float score = 0;
  int cls = 0, idx = 0, id = 0, frame_id = 0, start_frame = 0;
  bool confirmed = false;
  bool was_lost = false;         // scratch: lost before this frame
  std::vector<Obs> obs;
  // with_reid (`model: auto`): the detection's normalised appearance vector; a track's 0.9-EMA of them, re-normalised (BOTrack's rule)
  std::vector<float> curr_feat, smooth_feat;
  void update_features(const std::vector<float>& f) {
    curr_feat = f;
    if (smooth_feat.empty()) { smooth_feat = f; }
    else for (size_t i = 0; i < f.size(); ++i) smooth_feat[i] = 0.9f * smooth_feat[i] + (1.f - 0.9f) * f[i];
    float ss = 0.f;
    for (float v : smooth_feat) ss += v * v;
    const float nrm = std::sqrt(ss);
    for (float& v : smooth_feat) v /= nrm;
  }
  void xyxy(double o[4]) const {
    o[0] = mean[0] - mean[2] / 2; o[1] = mean[1] - mean[3] / 2; o[2] = mean[0] + mean[2] / 2; o[3] = mean[1] + mean[3] / 2;
  }
};

double iou_pair(const double* a, const double* b) {
  const double iw = std::min(a[2], b[2]) - std::max(a[0], b[0]), ih = std::min(a[3], b[3]) - std::max(a[1], b[1]);
  if (iw <= 0 || ih <= 0) return 0.0;
  const double inter = iw * ih;
  return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter);
}
void to_z(const double* b, double z[4]) { z[0] = (b[0] + b[2]) / 2; z[1] = (b[1] + b[3]) / 2; z[2] = b[2] - b[0]; z[3] = b[3] - b[1]; }
}  // namespace

struct TrackTrackTracker::Impl {
  gtx_tracker_config cfg;
  trk::Kalman kf{true};
  std::vector<TTrack> tracked, lost;
  int frame_id = 0, next_id = 0, max_time_lost = 30;
  static constexpr int kDeltaT = 3;

  double angle(const TTrack& t, const double* det) const {
    if (t.obs.size() < 2) return 0.0;
    const Obs& last = t.obs.back();
    const double* prev = nullptr;
    for (int dt = kDeltaT; dt > 0 && !prev; --dt)
      for (const Obs& o : t.obs)
        if (o.frame == last.frame - dt) { prev = o.b; break; }
    if (!prev) prev = t.obs[t.obs.size() - 2].b;
    static const int cx[4] = {0, 2, 0, 2}, cy[4] = {1, 1, 3, 3};
    double tot = 0.0;
    for (int k = 0; k < 4; ++k) {
      const double vx = last.b[cx[k]] - prev[cx[k]], vy = last.b[cy[k]] - prev[cy[k]];
      const double ux = det[cx[k]] - prev[cx[k]], uy = det[cy[k]] - prev[cy[k]];
      const double nv = std::hypot(vx, vy), nu = std::hypot(ux, uy);
      if (nv < 1e-9 || nu < 1e-9) continue;
      const double c = std::min(1.0, std::max(-1.0, (vx * ux + vy * uy) / (nv * nu)));
      tot += std::acos(c) / M_PI;
    }
    return tot / 4.0;
  }

  // all mutually-minimal pairs below the threshold at once, rows and columns removed, threshold lowered; until none qualifies
  void iterate(std::vector<double> C, int n, int m, double thr, std::vector<std::pair<int, int>>& out) const {
    out.clear();
    std::vector<int> rmin(n), cmin(m);
    while (n > 0 && m > 0 && thr > 0) {
      for (int i = 0; i < n; ++i) {
        int b = 0;
        for (int j = 1; j < m; ++j)
          if (C[(size_t)i * m + j] < C[(size_t)i * m + b]) b = j;
        rmin[i] = b;
      }
      for (int j = 0; j < m; ++j) {
        int b = 0;
        for (int i = 1; i < n; ++i)
          if (C[(size_t)i * m + j] < C[(size_t)b * m + j]) b = i;
        cmin[j] = b;
      }
      std::vector<std::pair<int, int>> pairs;
      for (int i = 0; i < n; ++i) {
        const double v = C[(size_t)i * m + rmin[i]];
        if (std::isfinite(v) && cmin[rmin[i]] == i && v < thr) pairs.emplace_back(i, rmin[i]);
      }
      if (pairs.empty()) break;
      for (const auto& p : pairs) {
        out.push_back(p);
        for (int j = 0; j < m; ++j) C[(size_t)p.first * m + j] = kInf;
        for (int i = 0; i < n; ++i) C[(size_t)i * m + p.second] = kInf;
      }
      thr -= (double)cfg.reduce_step;
    }
  }

  void absorb(TTrack& t, const TTrack& d) {
    double z[4];
    to_z(d.det, z);
    kf.update(t.mean, t.cov, z);
    t.state = kTracked;
    t.frame_id = frame_id;
    t.score = d.score; t.cls = d.cls; t.idx = d.idx;
    Obs o;
    o.frame = frame_id;
    std::memcpy(o.b, d.det, sizeof o.b);
    t.obs.push_back(o);
    if (t.obs.size() > 64) t.obs.erase(t.obs.begin(), t.obs.end() - 64);
    if ((int)t.obs.size() >= cfg.min_track_len) t.confirmed = true;
    if (!d.curr_feat.empty()) t.update_features(d.curr_feat);
  }
};

TrackTrackTracker::TrackTrackTracker(const gtx_tracker_config& cfg) : impl_(new Impl) {
  GTX_CHECK(cfg.type == 5, "tracker type %d: tracktrack (5) lives here", cfg.type);
  impl_->cfg = cfg;
  const int fr = cfg.frame_rate > 0 ? cfg.frame_rate : 30;
  impl_->max_time_lost = (int)(fr / 30.0 * cfg.track_buffer);
}
TrackTrackTracker::~TrackTrackTracker() = default;

void TrackTrackTracker::reset() {
  impl_->tracked.clear();
  impl_->lost.clear();
  impl_->frame_id = 0;
  impl_->next_id = 0;
}

void TrackTrackTracker::update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap, int* n_out,
                               float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx, const float* feats, int feat_dim) {
  Impl& S = *impl_;
  const gtx_tracker_config& A = S.cfg;
  const bool reid = A.with_reid != 0;
  GTX_CHECK(!reid || n == 0 || (feats != nullptr && feat_dim > 0), "tracktrack: with_reid needs an appearance vector per detection (gtx_tracker_update_feats)");
  S.frame_id += 1;
  std::vector<TTrack> dets;
  std::vector<char> low;
  for (int i = 0; i < n; ++i) {
    if (!(conf[i] > A.track_low_thresh)) continue;
    TTrack d;
    for (int k = 0; k < 4; ++k) d.det[k] = (double)xyxy[4 * i + k];
    d.score = conf[i]; d.cls = cls[i]; d.idx = i;
    if (reid) {                                   // feat / ||feat|| in float32
      d.curr_feat.assign(feats + (size_t)i * feat_dim, feats + (size_t)(i + 1) * feat_dim);
      float ss = 0.f;
      for (float v : d.curr_feat) ss += v * v;
      const float nrm = std::sqrt(ss);
      for (float& v : d.curr_feat) v /= nrm;
      d.smooth_feat = d.curr_feat;
    }
    dets.push_back(d);
    low.push_back(conf[i] < A.track_high_thresh ? 1 : 0);
  }
  // pool = tracked, then lost
  std::vector<TTrack> pool;
  pool.reserve(S.tracked.size() + S.lost.size());
  for (TTrack& t : S.tracked) { t.was_lost = false; pool.push_back(std::move(t)); }
  for (TTrack& t : S.lost) { t.was_lost = true; pool.push_back(std::move(t)); }
  for (TTrack& t : pool) {
    if (t.state != kTracked) { t.mean[6] = 0; t.mean[7] = 0; }
    S.kf.predict(t.mean, t.cov);
  }
  if (gmc) {                                      // BOTSORT.multi_gmc: mean <- kron(I4, R) mean (+t on xy), cov <- R8 cov R8^T
    const double R[4] = {gmc[0], gmc[1], gmc[3], gmc[4]}, tx = gmc[2], ty = gmc[5];
    for (TTrack& t : pool) {
      double m[8];
      for (int b = 0; b < 4; ++b) {
        m[2 * b] = R[0] * t.mean[2 * b] + R[1] * t.mean[2 * b + 1];
        m[2 * b + 1] = R[2] * t.mean[2 * b] + R[3] * t.mean[2 * b + 1];
      }
      m[0] += tx; m[1] += ty;
      std::memcpy(t.mean, m, sizeof m);
      double tmp[64], out[64];
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
          const int bi = i / 2, ri = i % 2;
          tmp[i * 8 + j] = R[ri * 2] * t.cov[(2 * bi) * 8 + j] + R[ri * 2 + 1] * t.cov[(2 * bi + 1) * 8 + j];
        }
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
          const int bj = j / 2, rj = j % 2;
          out[i * 8 + j] = tmp[i * 8 + 2 * bj] * R[rj * 2] + tmp[i * 8 + 2 * bj + 1] * R[rj * 2 + 1];
        }
      std::memcpy(t.cov, out, sizeof out);
    }
  }

  // ---- the cost matrix: HMIoU (twice: no appearance model), confidence, corner angle; low-score detections pay penalty_p ----
  const int np_ = (int)pool.size(), nd = (int)dets.size();
  std::vector<double> C((size_t)np_ * nd, kInf);
  for (int i = 0; i < np_; ++i) {
    double a[4];
    pool[i].xyxy(a);
    for (int j = 0; j < nd; ++j) {
      const double* b = dets[j].det;
      const double iou = iou_pair(a, b);
      if (iou <= 0.0) continue;
      const double hi = (std::min(a[3], b[3]) - std::max(a[1], b[1])) / (std::max(a[3], b[3]) - std::min(a[1], b[1]));
      const double dist = 1.0 - iou * hi;
      double app = dist;                          // no appearance model: the HMIoU distance once more (default.yaml:456)
      if (reid && !pool[i].smooth_feat.empty()) {  // cosine distance between the track's smoothed vector and the detection's, in [0, 1]
        float dot = 0.f;
        const std::vector<float>&u = pool[i].smooth_feat, &w = dets[j].curr_feat;
        for (size_t k = 0; k < u.size(); ++k) dot += u[k] * w[k];
        app = std::min(1.0, std::max(0.0, 1.0 - (double)dot));
      }
      const double c = (double)A.iou_weight * dist + (double)A.reid_weight * app +
                       (double)A.conf_weight * std::fabs((double)pool[i].score - (double)dets[j].score) + (double)A.angle_weight * S.angle(pool[i], b);
      C[(size_t)i * nd + j] = c + (low[j] ? (double)A.penalty_p : 0.0);
    }
  }
  std::vector<std::pair<int, int>> matches;
  S.iterate(C, np_, nd, (double)A.match_thresh, matches);
  std::vector<char> mt(np_, 0), md(nd, 0);
  for (const auto& p : matches) { S.absorb(pool[p.first], dets[p.second]); mt[p.first] = 1; md[p.second] = 1; }
  if (A.lost_match_thr > 0.f) {                   // tracks lost before this frame, once more against the unmatched high-score detections
    std::vector<int> rows, cols;
    for (int i = 0; i < np_; ++i) if (!mt[i] && pool[i].was_lost) rows.push_back(i);
    for (int j = 0; j < nd; ++j) if (!md[j] && !low[j]) cols.push_back(j);
    if (!rows.empty() && !cols.empty()) {
      std::vector<double> sub(rows.size() * cols.size());
      for (size_t a = 0; a < rows.size(); ++a)
        for (size_t b = 0; b < cols.size(); ++b) sub[a * cols.size() + b] = C[(size_t)rows[a] * nd + cols[b]];
      std::vector<std::pair<int, int>> m2;
      S.iterate(sub, (int)rows.size(), (int)cols.size(), (double)A.lost_match_thr, m2);
      for (const auto& p : m2) { S.absorb(pool[rows[p.first]], dets[cols[p.second]]); mt[rows[p.first]] = 1; md[cols[p.second]] = 1; }
    }
  }
  std::vector<std::array<double, 4>> active;
  for (int i = 0; i < np_; ++i) {
    if (mt[i]) { std::array<double, 4> a; pool[i].xyxy(a.data()); active.push_back(a); continue; }
    if (pool[i].was_lost) continue;
    pool[i].state = pool[i].confirmed ? kLost : kRemoved;
  }
  // ---- track-aware initialisation ----
  std::vector<TTrack> born;
  for (int j = 0; j < nd; ++j) {
    TTrack& d = dets[j];
    if (md[j] || low[j] || d.score < A.new_track_thresh) continue;
    bool drop = false;
    for (const auto& a : active) if (iou_pair(d.det, a.data()) > (double)A.tai_thr) { drop = true; break; }
    if (!drop) for (const TTrack& b : born) if (iou_pair(d.det, b.det) > (double)A.tai_thr) { drop = true; break; }
    if (drop) continue;
    d.id = ++S.next_id;
    double z[4];
    to_z(d.det, z);
    S.kf.initiate(z, d.mean, d.cov);
    d.has_mean = true;
    d.state = kTracked;
    d.frame_id = d.start_frame = S.frame_id;
    Obs o;
    o.frame = S.frame_id;
    std::memcpy(o.b, d.det, sizeof o.b);
    d.obs.assign(1, o);
    d.confirmed = S.frame_id == 1 || A.min_track_len <= 1;
    born.push_back(d);
  }
  for (TTrack& t : pool)
    if (t.was_lost && t.state == kLost && S.frame_id - t.frame_id > S.max_time_lost) t.state = kRemoved;
  S.tracked.clear();
  S.lost.clear();
  for (TTrack& t : pool) {
    if (t.state == kTracked) S.tracked.push_back(std::move(t));
    else if (t.state == kLost) S.lost.push_back(std::move(t));
  }
  for (TTrack& t : born) S.tracked.push_back(std::move(t));
  int k = 0;
  for (const TTrack& t : S.tracked) {
    if (!t.confirmed || t.frame_id != S.frame_id) continue;
    GTX_CHECK(k < cap, "tracker: more than %d active tracks", cap);
    double b[4];
    t.xyxy(b);
    for (int q = 0; q < 4; ++q) out_xyxy[4 * k + q] = (float)b[q];
    out_id[k] = t.id; out_score[k] = t.score; out_cls[k] = t.cls; out_det_idx[k] = t.idx;
    ++k;
  }
  *n_out = k;
}

}  // namespace gtx
