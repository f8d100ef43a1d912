#include "geometry.hpp"

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

}  // namespace gtx
