// The "pair format" of the default fp32 path (DT_F32S): an fp32-grade value x is carried as two fp16 numbers,
//     hi = fp16(clamp(x, +-65504)),   lo = fp16(x - hi)       (x - hi is exact in fp32; hi + lo is exact in fp32),
// i.e. 22 significand bits, and an NHWC tensor keeps its 4 bytes per element: the 32 bytes of every aligned group of 8
// channels hold the group's 8 hi halves followed by its 8 lo halves. A convolution stages a group as two 16-byte chunks
// that go into its LDS row unchanged (conv_igemm_split.hip); producers (stem, conv epilogue) split once.
// Channel counts, strides and slice offsets of pair tensors are multiples of 8. Addresses keep "element" units: element e
// of a buffer starts the group at byte 4 * (e & ~7), its hi half is at + 2 * (e & 7), its lo half at + 16 + 2 * (e & 7).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace gtx {

// Host-side conversion of whole buffers (n a multiple of 8), used where plain fp32 host arrays meet the library
// (gtx_op_conv2d, gtx_op_sppf_pool, gtx_detector_layer_output). Returns true when a value was clamped.
inline bool f32_to_pairs(const float* src, void* dst, size_t n) {
  bool sat = false;
  uint8_t* d = static_cast<uint8_t*>(dst);
  for (size_t g = 0; g + 8 <= n; g += 8)
    for (int e = 0; e < 8; ++e) {
      float x = src[g + e];
      const float c = x > 65504.f ? 65504.f : (x < -65504.f ? -65504.f : x);
      sat |= !(c == x);
      x = (c == c) ? c : -65504.f;                 // NaN: what v_med3_f32 returns on the device
      const _Float16 hi = (_Float16)x;
      const _Float16 lo = (_Float16)(x - (float)hi);
      memcpy(d + g * 4 + 2 * e, &hi, 2);
      memcpy(d + g * 4 + 16 + 2 * e, &lo, 2);
    }
  return sat;
}

inline float pair_element(const void* base, size_t e) {
  const uint8_t* b = static_cast<const uint8_t*>(base) + (e & ~(size_t)7) * 4;
  _Float16 hi, lo;
  memcpy(&hi, b + 2 * (e & 7), 2);
  memcpy(&lo, b + 16 + 2 * (e & 7), 2);
  return (float)hi + (float)lo;
}

inline void pairs_to_f32(const void* src, float* dst, size_t n) {
  for (size_t e = 0; e < n; ++e) dst[e] = pair_element(src, e);
}

}  // namespace gtx
