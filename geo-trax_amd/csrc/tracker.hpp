// Host-side multi-object tracker (ByteTrack / BoT-SORT association), C++.
#pragma once
#include <memory>
#include <vector>

#include "../../include/gtx.h"
#include "common.hpp"

namespace gtx {
class ByteTracker {
 public:
  explicit ByteTracker(const gtx_tracker_config& cfg);
  ~ByteTracker();
  void reset();
  // feats (may be null): [n][feat_dim] appearance vectors, used when gtx_tracker_config.with_reid is set (BoT-SORT)
  void update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap, int* n_out,
              float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx,
              const float* feats = nullptr, int feat_dim = 0);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
// OC-SORT (tracker.ocsort of the reference's config, default.yaml:391-404): ocsort.cpp. Same update() contract.
class OcSortTracker {
 public:
  explicit OcSortTracker(const gtx_tracker_config& cfg);
  ~OcSortTracker();
  void reset();
  void update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap, int* n_out,
              float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx, const float* feats = nullptr, int feat_dim = 0);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// TrackTrack (tracker.tracktrack, default.yaml:445-470): tracktrack.cpp. Same update() contract.
class TrackTrackTracker {
 public:
  explicit TrackTrackTracker(const gtx_tracker_config& cfg);
  ~TrackTrackTracker();
  void reset();
  void update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap, int* n_out,
              float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx, const float* feats = nullptr, int feat_dim = 0);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// Minimum-cost assignment of a dense rows x cols matrix that matches min(rows, cols) pairs (what
// scipy.optimize.linear_sum_assignment / lap.lapjv(extend_cost=True) return). x[r] = column of row r or -1.
void lap_full(const std::vector<double>& cost, int rows, int cols, std::vector<int>& x);
// lap.lapjv(cost, extend_cost=True, cost_limit=limit) as ByteTrack / BoT-SORT call it: a pair is matched only below `limit`,
// leaving a row or a column unmatched costs limit / 2 each. x[r] = column or -1, y[c] = row or -1 (tracker.cpp's sparse solver).
void lap_limited(const float* cost, int rows, int cols, double limit, std::vector<int>& x, std::vector<int>& y);
}  // namespace gtx

struct gtx_tracker {
  std::unique_ptr<gtx::ByteTracker> impl;
  std::unique_ptr<gtx::OcSortTracker> oc;
  std::unique_ptr<gtx::TrackTrackTracker> tt;
};
