// The constant-velocity Kalman filter of ultralytics/trackers/utils/kalman_filter.py (KalmanFilterXYAH for ByteTrack, KalmanFilterXYWH
// for BoT-SORT), float64 like the numpy original. Shared by tracker.cpp and tracktrack.cpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

namespace gtx {
namespace trk {

struct Kalman {
  bool xywh;  // false: XYAH (ByteTrack), true: XYWH (BoT-SORT)
  static constexpr double swp = 1.0 / 20, swv = 1.0 / 160;

  void stds(const double* m, double k_pos, double k_vel, double a_pos, double a_vel, double* s) const {
    if (xywh) {
      const double w = m[2], h = m[3];
      s[0] = k_pos * swp * w; s[1] = k_pos * swp * h; s[2] = k_pos * swp * w; s[3] = k_pos * swp * h;
      s[4] = k_vel * swv * w; s[5] = k_vel * swv * h; s[6] = k_vel * swv * w; s[7] = k_vel * swv * h;
    } else {
      const double h = m[3];
      s[0] = k_pos * swp * h; s[1] = k_pos * swp * h; s[2] = a_pos; s[3] = k_pos * swp * h;
      s[4] = k_vel * swv * h; s[5] = k_vel * swv * h; s[6] = a_vel; s[7] = k_vel * swv * h;
    }
  }

  void initiate(const double z[4], double* mean, double* cov) const {
    for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0; }
    double s[8];
    stds(mean, 2, 10, 1e-2, 1e-5, s);
    std::fill(cov, cov + 64, 0.0);
    for (int i = 0; i < 8; ++i) cov[i * 9] = s[i] * s[i];
  }

  // mean <- F mean, cov <- F cov F^T + Q, F = I + shift(4)
  void predict(double* mean, double* cov) const {
    double s[8];
    stds(mean, 1, 1, 1e-2, 1e-5, s);
    for (int i = 0; i < 4; ++i) mean[i] += mean[4 + i];
    double t[64];
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) t[i * 8 + j] = cov[i * 8 + j] + (i < 4 ? cov[(i + 4) * 8 + j] : 0.0);  // F cov
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) cov[i * 8 + j] = t[i * 8 + j] + (j < 4 ? t[i * 8 + j + 4] : 0.0);      // (.) F^T
    for (int i = 0; i < 8; ++i) cov[i * 9] += s[i] * s[i];
  }

  // Measurement update with H = [I4 0]: S = P[:4,:4] + R, K = P[:, :4] S^-1, mean += K (z - mean[:4]),
  // P -= K S K^T = P[:, :4] K^T ... computed as P - K P[:4, :]. S^-1 comes from one Cholesky factorisation
  // (4 square roots, 4 reciprocals); everything else is multiply-add.
  void update(double* mean, double* cov, const double z[4]) const {
    double s[8];
    stds(mean, 1, 1, 1e-1, 0, s);
    double S[16];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) S[i * 4 + j] = cov[i * 8 + j] + (i == j ? s[i] * s[i] : 0.0);
    // Cholesky S = L L^T, then Linv = L^-1 (lower triangular), Sinv = Linv^T Linv
    double L[16] = {0}, rd[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j <= i; ++j) {
        double v = S[i * 4 + j];
        for (int k = 0; k < j; ++k) v -= L[i * 4 + k] * L[j * 4 + k];
        if (i == j) { L[i * 4 + i] = std::sqrt(v); rd[i] = 1.0 / L[i * 4 + i]; }
        else L[i * 4 + j] = v * rd[j];
      }
    double Li[16] = {0};
    for (int c = 0; c < 4; ++c) {
      Li[c * 4 + c] = rd[c];
      for (int r = c + 1; r < 4; ++r) {
        double v = 0;
        for (int k = c; k < r; ++k) v -= L[r * 4 + k] * Li[k * 4 + c];
        Li[r * 4 + c] = v * rd[r];
      }
    }
    double Si[16];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j <= i; ++j) {
        double v = 0;
        for (int k = i; k < 4; ++k) v += Li[k * 4 + i] * Li[k * 4 + j];
        Si[i * 4 + j] = Si[j * 4 + i] = v;
      }
    double K[32];
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 4; ++c) {
        double v = 0;
        for (int k = 0; k < 4; ++k) v += cov[r * 8 + k] * Si[k * 4 + c];
        K[r * 4 + c] = v;
      }
    double innov[4];
    for (int i = 0; i < 4; ++i) innov[i] = z[i] - mean[i];
    for (int r = 0; r < 8; ++r)
      for (int i = 0; i < 4; ++i) mean[r] += innov[i] * K[r * 4 + i];
    double top[32];                                  // P[:4, :] before it is overwritten
    std::memcpy(top, cov, sizeof top);
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 8; ++c) {
        double v = 0;
        for (int k = 0; k < 4; ++k) v += K[r * 4 + k] * top[k * 8 + c];
        cov[r * 8 + c] -= v;
      }
  }
};

}  // namespace trk
}  // namespace gtx
