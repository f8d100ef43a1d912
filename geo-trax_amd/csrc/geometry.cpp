#include "geometry.hpp"

#include <vector>

#include <algorithm>
#include <cmath>
#include <limits>

namespace gtx {

namespace {
inline void project(const double H[9], double x, double y, double& ox, double& oy) {
  // Same evaluation order as OpenCV's perspectiveTransform: reciprocal of w, then multiply.
  double w = x * H[6] + y * H[7] + H[8];
  if (std::fabs(w) > std::numeric_limits<double>::epsilon()) {
    w = 1.0 / w;
    ox = (x * H[0] + y * H[1] + H[2]) * w;
    oy = (x * H[3] + y * H[4] + H[5]) * w;
  } else {
    ox = oy = 0.0;
  }
}
}  // namespace

void warp_boxes(const double H[9], const float* in, int n, float* out) {
  for (int i = 0; i < n; ++i) {
    const double cx = in[4 * i + 0], cy = in[4 * i + 1], w = in[4 * i + 2], h = in[4 * i + 3];
    const double xs[4] = {cx - w / 2, cx + w / 2, cx + w / 2, cx - w / 2};
    const double ys[4] = {cy - h / 2, cy - h / 2, cy + h / 2, cy + h / 2};
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    for (int k = 0; k < 4; ++k) {
      double px, py;
      project(H, xs[k], ys[k], px, py);
      if (k == 0) { x0 = x1 = px; y0 = y1 = py; }
      else { x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py); }
    }
    out[4 * i + 0] = (float)((x0 + x1) / 2);
    out[4 * i + 1] = (float)((y0 + y1) / 2);
    out[4 * i + 2] = (float)(x1 - x0);
    out[4 * i + 3] = (float)(y1 - y0);
  }
}

void perspective_points(const double H[9], const double* x, const double* y, int n, double* ox, double* oy) {
  for (int i = 0; i < n; ++i) project(H, x[i], y[i], ox[i], oy[i]);
}

bool invert3x3(const double m[9], double inv[9]) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  if (det == 0.0 || !std::isfinite(det)) return false;
  const double r = 1.0 / det;
  inv[0] = A * r; inv[1] = -(b * i - c * h) * r; inv[2] = (b * f - c * e) * r;
  inv[3] = B * r; inv[4] = (a * i - c * g) * r;  inv[5] = -(a * f - c * d) * r;
  inv[6] = C * r; inv[7] = -(a * h - b * g) * r; inv[8] = (a * e - b * d) * r;
  return true;
}

namespace {
inline unsigned hash_u32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
}  // namespace

bool estimate_affine_partial(const float* p, const float* q, int n, unsigned seed, double A[6], int* n_inliers) {
  constexpr int kHyp = 512;
  constexpr double kThr2 = 3.0 * 3.0;
  if (n_inliers) *n_inliers = 0;
  if (n < 2) return false;
  auto P = [&](int i, int k) { return (double)p[2 * i + k]; };
  auto Q = [&](int i, int k) { return (double)q[2 * i + k]; };
  int best = -1;
  double M[6] = {1, 0, 0, 0, 1, 0};
  for (unsigned hyp = 0; hyp < (unsigned)kHyp; ++hyp) {
    const int i = (int)(hash_u32(seed ^ hash_u32(2u * hyp)) % (unsigned)n), j = (int)(hash_u32(seed ^ hash_u32(2u * hyp + 1u)) % (unsigned)n);
    if (i == j) continue;
    const double dpx = P(j, 0) - P(i, 0), dpy = P(j, 1) - P(i, 1), den = dpx * dpx + dpy * dpy;
    if (den < 1e-12) continue;
    const double dqx = Q(j, 0) - Q(i, 0), dqy = Q(j, 1) - Q(i, 1);
    const double a = (dpx * dqx + dpy * dqy) / den, b = (dpx * dqy - dpy * dqx) / den;
    const double tx = Q(i, 0) - (a * P(i, 0) - b * P(i, 1)), ty = Q(i, 1) - (b * P(i, 0) + a * P(i, 1));
    int cnt = 0;
    for (int k = 0; k < n; ++k) {
      const double ex = a * P(k, 0) - b * P(k, 1) + tx - Q(k, 0), ey = b * P(k, 0) + a * P(k, 1) + ty - Q(k, 1);
      cnt += (ex * ex + ey * ey < kThr2) ? 1 : 0;
    }
    if (cnt > best) { best = cnt; M[0] = a; M[1] = -b; M[2] = tx; M[3] = b; M[4] = a; M[5] = ty; }
  }
  if (best < 0) return false;
  int inl_n = best;
  for (int round = 0; round < 3; ++round) {
    double mpx = 0, mpy = 0, mqx = 0, mqy = 0;
    int cnt = 0;
    std::vector<char> inl(n, 0);
    for (int k = 0; k < n; ++k) {
      const double ex = M[0] * P(k, 0) + M[1] * P(k, 1) + M[2] - Q(k, 0), ey = M[3] * P(k, 0) + M[4] * P(k, 1) + M[5] - Q(k, 1);
      if (ex * ex + ey * ey < kThr2) { inl[k] = 1; ++cnt; mpx += P(k, 0); mpy += P(k, 1); mqx += Q(k, 0); mqy += Q(k, 1); }
    }
    inl_n = cnt;
    if (cnt < 2) break;
    mpx /= cnt; mpy /= cnt; mqx /= cnt; mqy /= cnt;
    double den = 0, sa = 0, sb = 0;
    for (int k = 0; k < n; ++k) {
      if (!inl[k]) continue;
      const double px = P(k, 0) - mpx, py = P(k, 1) - mpy, qx = Q(k, 0) - mqx, qy = Q(k, 1) - mqy;
      den += px * px + py * py;
      sa += px * qx + py * qy;
      sb += px * qy - py * qx;
    }
    if (den <= 1e-12) break;
    const double a = sa / den, b = sb / den;
    M[0] = a; M[1] = -b; M[2] = mqx - (a * mpx - b * mpy);
    M[3] = b; M[4] = a; M[5] = mqy - (b * mpx + a * mpy);
  }
  for (int k = 0; k < 6; ++k) A[k] = M[k];
  if (n_inliers) *n_inliers = inl_n;
  return true;
}

}  // namespace gtx
