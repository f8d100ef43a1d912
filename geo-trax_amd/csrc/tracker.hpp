// Host-side multi-object tracker (ByteTrack / BoT-SORT association), C++.
#pragma once
#include <memory>

#include "../../include/gtx.h"
#include "common.hpp"

namespace gtx {
class ByteTracker {
 public:
  explicit ByteTracker(const gtx_tracker_config& cfg);
  ~ByteTracker();
  void reset();
  void update(int n, const float* xyxy, const float* conf, const int* cls, const double* gmc, int cap, int* n_out,
              float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};
}  // namespace gtx

struct gtx_tracker {
  std::unique_ptr<gtx::ByteTracker> impl;
};
