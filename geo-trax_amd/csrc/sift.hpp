// SIFT / RootSIFT keypoints and descriptors on the GPU (sift.hip) -- the detector stage of the
// orthophoto / master-frame registration (reference: geotrax/utils/registration.py:59-85,
// detector_name='rsift').
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "common.hpp"

namespace gtx {

struct SiftKeypoint {
  float x, y;        // full-resolution pixels of the input image
  float size;        // diameter of the meaningful neighbourhood
  float angle;       // degrees, [0, 360)
  float response;    // |contrast|
  int octave;        // OpenCV packing: octave | layer << 8 | sub-layer offset << 16
};

class Sift {
 public:
  // Buffers are sized for images up to max_h x max_w (the Gaussian / DoG pyramids of the doubled
  // image stay resident: ~ 59 bytes per doubled pixel).
  Sift(int device, hipStream_t stream, int max_h, int max_w);
  ~Sift();
  // image: BGR u8 [h][w][3] on the HOST. Keypoints (OpenCV order: by octave, layer, row, column,
  // orientation bin) with their descriptors (128 floats each; RootSIFT when root) are left on the
  // device; n = number of keypoints (<= max_features, the strongest responses are retained).
  void detect_and_compute(const uint8_t* image_bgr, int h, int w, int max_features, bool root, float root_eps);
  int count() const;
  const float* descriptors_dev() const;         // [n][128] fp32
  const float2* positions_dev() const;          // [n] (x, y)
  void download(std::vector<SiftKeypoint>& kps, std::vector<float>& desc) const;
  const std::vector<SiftKeypoint>& keypoints_host() const;   // the keypoints alone: no copy of the descriptors (128 MB at 250 000 keypoints)
  // test hooks: pyramid image (kind 0 = Gaussian, 1 = DoG) of the last image
  void pyramid_image(int kind, int octave, int layer, std::vector<float>& out, int* h, int* w) const;
  int n_octaves() const;
  // GPU ms of the last detect_and_compute: [0] upload + gray + pyramid, [1] extrema + refine + orientation, [2] descriptors; [3] = pixels
  // of the doubled base image (the pyramid holds 11 x 4/3 fp32 images of that size)
  void stage_ms(float out[4]) const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace gtx
