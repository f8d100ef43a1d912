// Small projective-geometry helpers of the path (host, f64).
#pragma once
#include <cstdint>

struct gtx_ctx;
struct gtx_georef_chain;

namespace gtx {

// stabilo Stabilizer.transform_cur_boxes() as pinned on the reference's golden output
// (SURVEY.md K10): the four corners of each xywh box go through H, the axis-aligned bounding
// rectangle of the four images comes back as xywh.
void warp_boxes(const double H[9], const float* xywh_in, int n, float* xywh_out);

// cv2.perspectiveTransform on f64 points (geotrax/georeference.py:599-605).
void perspective_points(const double H[9], const double* x, const double* y, int n, double* ox, double* oy);

// The georeference stage's per-row chain pixel -> orthophoto pixel -> lat/lon -> projected metres
// (geotrax/georeference.py:173-177, 599-628). Device kernel in georef.hip.
void georef_points(gtx_ctx* ctx, const gtx_georef_chain& chain, const double* x, const double* y, int n, double* ox, double* oy,
                   double* lat, double* lon, double* east, double* north);

// cv2.warpPerspective(frame, H, (w, h)): dst(x,y) = bilinear src(H^-1 (x,y)), constant 0 border
// (geotrax/visualize.py:289). Device kernel in warp.hip.
void warp_frame(gtx_ctx* ctx, const uint8_t* src_bgr, int h, int w, const double H[9], uint8_t* dst_bgr);
// yuv.hip: I420 frame in HBM -> BGR frame in HBM (BT.601 limited range, OpenCV's fixed point), asynchronous
void yuv420_to_bgr_dev(gtx_ctx* ctx, const void* yuv, int h, int w, void* bgr);
void warp_frame_dev(gtx_ctx* ctx, const void* src_bgr, int h, int w, const double H[9], void* dst_bgr);   // both in HBM, asynchronous

// cv2.estimateAffinePartial2D(prev, cur, RANSAC) as ultralytics' GMC calls it for the feature-based methods (`gmc_method: orb` /
// `sift`, default.yaml:374): a 4-parameter similarity p -> q from matched points. 512 two-point hypotheses (counter-hash sampling),
// the one with the most points within 3 px wins (first on ties), three rounds of least squares on the inliers: the procedure of
// gmc.hip's sparseOptFlow fit, on the host (a few hundred matches). p, q: [n][2] float32. Returns false when no model exists.
bool estimate_affine_partial(const float* p, const float* q, int n, unsigned seed, double A[6], int* n_inliers);

// 3x3 inverse (adjugate / det). Returns false if singular.
bool invert3x3(const double m[9], double inv[9]);

}  // namespace gtx
