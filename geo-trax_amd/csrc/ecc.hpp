// Global motion compensation, method 'ecc' (gmc_method: ecc, geotrax/cfg/default.yaml:374,419,467): ultralytics'
// GMC.apply_ecc = cv2.findTransformECC(first frame, current frame, MOTION_EUCLIDEAN, 5000 iterations / 1e-6) on the
// Gaussian-blurred half-resolution gray image. gfx950 only; kernels in ecc.hip. The procedure is oracle/ecc_ref.py's,
// operation by operation (that file says which OpenCV source it restates and which two upstream properties it keeps).
#pragma once
#include <memory>

#include "common.hpp"

namespace gtx {

class Ecc {
 public:
  Ecc(int device, hipStream_t stream, int frame_h, int frame_w, int max_iters, double eps);
  ~Ecc();
  void reset();                                               // forget the template (nothing may be in flight)
  // false (default): the template stays the first frame of the sequence, as upstream's apply_ecc leaves `prevFrame`; true: every
  // collected frame becomes the template of the next one (frame-to-frame warps: what the method's name promises)
  void set_replace_template(bool on);
  // true (default): warpAffine's bilinear path as OpenCV >= 4.11 computes it (source positions in floating point); false: as through
  // 4.10 (fixed point, 1/32 pixel) -- under which many fits dither in the coefficient's sixth decimal and run to the iteration cap
  void set_exact_positions(bool on);
  // Queue a BGR u8 frame that lives in HBM: its blurred half-resolution image is made NOW, on `producer` (the stream that
  // wrote the frame, e.g. a detector's: the frame buffer may be reused by later work on that stream), into the next slot of a
  // 32-deep ring. collect() returns the frames in submission order. One thread may submit while another collects.
  void submit_frame_dev(const void* frame_bgr_dptr, int h, int w, hipStream_t producer);
  void submit_frame(const uint8_t* frame_bgr, int h, int w);  // host frame: uploaded on the object's own stream
  // A: row-major 2x3 (float32 values), template -> frame in HALF-resolution pixels, identity for the first frame.
  // info = {iterations run, status: 0 finished, 1 NaN correlation, 2 correlation about to be minimised}; rho: the last coefficient
  void collect(double A[6], int info[2], double* rho);
  // test hook: which 0 = the image of the frame collected last, 1 = the template; out [frame_h / 2][frame_w / 2]
  void debug_image(int which, float* out) const;
  int pending() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace gtx
