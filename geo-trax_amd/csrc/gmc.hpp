// BoT-SORT global motion compensation ('sparseOptFlow') on the GPU, see gmc.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>

#include "common.hpp"

namespace gtx {

class Gmc {
 public:
  // gray_h x gray_w: the half-resolution gray image the method works on (frame size / 2).
  Gmc(int device, hipStream_t stream, int gray_h, int gray_w, int seed);
  ~Gmc();
  void reset();                                               // forget the previous frame (nothing may be in flight)
  void restart();                                             // the next submitted frame opens a new sequence; frames may be in flight
  // asynchronous pair: corners + flow against the previous frame + RANSAC on the stream / refit on the host.
  // Up to 64 frames may be submitted ahead; collect() returns them in submission order. One thread may submit
  // while another collects.
  void submit_gray_dev(const void* gray, int gh, int gw);
  // BGR u8 host frame [2*gray_h][2*gray_w][3]: gray + 2x2 mean on the GPU, then as above
  void submit_frame(const uint8_t* frame_bgr, int h, int w);
  // The same for a frame that already lives in HBM, queued like submit_gray_dev. restart: this frame opens a new
  // sequence (its warp is the identity; the next frame is compensated against it) -- a shard rank uses it to
  // hand the GMC the frame that precedes its batch in the clip.
  void submit_frame_dev(const void* frame_bgr_dptr, int h, int w, bool restart);
  // A: row-major 2x3 f64 in full-resolution pixels (identity on the first frame or when fewer than 5
  // points were tracked; valid tells which). stats = {corners of the previous frame, tracked, inliers}.
  void collect(double A[6], int* valid, int stats[3]);
  // test hook: which 0 = corners of the last frame, 1 = corners of the frame before, 2 = their LK positions
  void debug_points(int which, int cap, int* n, float* xy, int* status) const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace gtx
