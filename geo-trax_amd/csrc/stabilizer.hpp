// GPU homography stabilizer (ORB-style keypoints + Hamming matching + RANSAC homography).
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "../../include/gtx.h"
#include "common.hpp"

struct gtx_ctx;

namespace gtx {
class Stabilizer {
 public:
  Stabilizer(gtx_ctx* ctx, const gtx_stab_config& cfg);
  ~Stabilizer();
  void set_ref_frame(const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n);
  void set_ref_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n);
  void stabilize(const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n, double H[9], int* valid, int stats[4]);
  void stabilize_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n, double H[9], int* valid, int stats[4]);
  // asynchronous pair: enqueue the whole pass for a gray image in HBM / wait for it and refit
  void submit_gray_dev(const void* gray, int gh, int gw, const float* boxes_xywh, int n);
  void collect(double H[9], int* valid, int stats[4]);
  void keypoints(int which, int cap, int* n, float* xy, int* level, int* angle_bin, uint8_t* desc);
  void promote_cur_to_ref();      // the last stabilized frame's features become the reference (ref_multiplier 1 only)
  void matches(int cap, int* n, int* cur_idx, int* ref_idx, int* dist);
  // rotated sampling pattern table [256 bins][256 tests][ax, ay, bx, by] int8 (data, for the oracle)
  void pattern(int8_t* out) const;
  // GPU time (ms, stream-ordered events) of the last collected submit_gray_dev pass: keypoints -> matching -> RANSAC
  float last_ms() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// The steered-BRIEF sampling table [256 bins][256 tests][ax, ay, bx, by] (host only, no device needed).
void stabilizer_pattern_table(std::vector<int8_t>& out);

// clahe.hip: cv2.createCLAHE(2.0, (8, 8)).apply on a u8 image in HBM (src == dst allowed); luts: kClaheLutBytes of scratch.
constexpr int kClaheLutBytes = 8 * 8 * 256;
void clahe_dev(const uint8_t* src, int h, int w, uint8_t* luts, uint8_t* dst, hipStream_t s);
// cv2.createCLAHE(2.0, (8, 8)).apply(gray), host image in / out (what `clahe: true` runs on the working gray image).
void clahe_image(gtx_ctx* ctx, const uint8_t* gray, int h, int w, uint8_t* out);

// Robust homography (MSAC hypotheses on the GPU + IRLS refit on the host, f64) from n_match point pairs
// (x, y) -> (z, w) in HBM; threshold in pixels of the destination. false: no model.
bool ransac_homography(int device, hipStream_t s, const float4* d_pts, int n_match, unsigned seed, int n_hyp, int frame_w, int frame_h,
                       float threshold, double H[9], int* n_inliers);
}  // namespace gtx

struct gtx_stabilizer {
  std::unique_ptr<gtx::Stabilizer> impl;
};
