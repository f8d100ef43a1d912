// Image-to-image registration: RootSIFT keypoints (sift.hip) -> MFMA 2-NN matching (match_l2.hip)
// -> Lowe ratio -> robust homography (stabilizer.hip). Mirrors what
// geotrax/utils/registration.py:59-85 asks stabilo for.
#pragma once
#include "../../include/gtx.h"
#include "common.hpp"

struct gtx_ctx;

namespace gtx {
// H maps src pixels to dst pixels. stats = {n_src_keypoints, n_dst_keypoints, n_good_matches, n_inliers}.
// valid = 0 when fewer than 4 good matches or no model.
void register_images(gtx_ctx* ctx, const gtx_reg_config& cfg, const uint8_t* src_bgr, int sh, int sw, const uint8_t* dst_bgr, int dh,
                     int dw, double H[9], int* valid, int stats[4], float timings_ms[4]);
}  // namespace gtx
