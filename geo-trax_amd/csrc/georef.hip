// The per-row transform chain of the georeference stage in one pass:
//   frame pixel --H--> orthophoto pixel --affine geotransform--> latitude/longitude --transverse Mercator--> metres
// (geotrax/georeference.py:173-177 calling apply_homography :599-605, ortho2geo :608-615, geo2local :618-628).
// f64 throughout; one thread per point, structure-of-arrays in and out, so every access is a coalesced 8-B
// stream: 16 B in, up to 48 B out per point -- HBM-bound by construction, the ~40 f64 transcendental calls per
// point hide under it. The projection is the Krueger series to 6th order (sub-millimetre inside a zone); the
// reference gets the same numbers from pyproj.
#include <hip/hip_runtime.h>

#include <cmath>

#include "detector.hpp"
#include "geometry.hpp"

namespace gtx {

namespace {
struct ChainDev {
  double H[9];
  double ortho[6];        // lng0, lat0, dlng, dlat, skew_x, skew_y
  int projected;
  double lon0_rad, e, A, k0, fe, fn;
  double alpha[6];
};

__global__ __launch_bounds__(256) void georef_points_kernel(const ChainDev c, const double* __restrict__ x, const double* __restrict__ y,
                                                            int n, double* __restrict__ ox, double* __restrict__ oy,
                                                            double* __restrict__ lat, double* __restrict__ lon,
                                                            double* __restrict__ east, double* __restrict__ north) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double px = x[i], py = y[i];
  // cv2.perspectiveTransform: w == 0 maps to (0, 0)
  const double w = c.H[6] * px + c.H[7] * py + c.H[8];
  const double s = fabs(w) > 2.220446049250313e-16 ? 1.0 / w : 0.0;
  const double u = (c.H[0] * px + c.H[1] * py + c.H[2]) * s, v = (c.H[3] * px + c.H[4] * py + c.H[5]) * s;
  if (ox) ox[i] = u;
  if (oy) oy[i] = v;
  const double la = c.ortho[1] + c.ortho[3] * v + c.ortho[5] * u, lo = c.ortho[0] + c.ortho[2] * u + c.ortho[4] * v;
  if (lat) lat[i] = la;
  if (lon) lon[i] = lo;
  if (!c.projected) return;
  constexpr double kDeg = 0.017453292519943295;
  const double phi = la * kDeg, lam = lo * kDeg - c.lon0_rad;
  const double sp = sin(phi);
  const double t = sinh(atanh(sp) - c.e * atanh(c.e * sp));      // tangent of the conformal latitude
  const double cl = cos(lam);
  const double xi_p = atan2(t, cl), eta_p = asinh(sin(lam) / hypot(t, cl));
  double xi = xi_p, eta = eta_p;
#pragma unroll
  for (int j = 1; j <= 6; ++j) {
    xi += c.alpha[j - 1] * sin(2 * j * xi_p) * cosh(2 * j * eta_p);
    eta += c.alpha[j - 1] * cos(2 * j * xi_p) * sinh(2 * j * eta_p);
  }
  if (east) east[i] = c.fe + c.k0 * (c.A * eta);
  if (north) north[i] = c.fn + c.k0 * (c.A * xi);
}
}  // namespace

void georef_points(gtx_ctx* ctx, const gtx_georef_chain& ch, const double* x, const double* y, int n, double* ox, double* oy,
                   double* lat, double* lon, double* east, double* north) {
  GTX_CHECK(n >= 0, "georef_points: n = %d", n);
  if (n == 0) return;
  GTX_HIP(hipSetDevice(ctx->device));
  ChainDev c{};
  for (int i = 0; i < 9; ++i) c.H[i] = ch.H[i];
  for (int i = 0; i < 6; ++i) c.ortho[i] = ch.ortho[i];
  c.projected = ch.projected;
  if (ch.projected) {
    GTX_CHECK(ch.flattening > 0 && ch.flattening < 0.1 && ch.semi_major > 0, "georef_points: ellipsoid a=%g f=%g", ch.semi_major, ch.flattening);
    const double f = ch.flattening, nn = f / (2 - f);
    const double n2 = nn * nn, n3 = n2 * nn, n4 = n2 * n2, n5 = n4 * nn, n6 = n3 * n3;
    c.A = ch.semi_major / (1 + nn) * (1 + n2 / 4 + n4 / 64 + n6 / 256);
    c.alpha[0] = nn / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    c.alpha[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    c.alpha[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    c.alpha[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    c.alpha[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    c.alpha[5] = 212378941 * n6 / 319334400;
    c.e = std::sqrt(f * (2 - f));
    c.lon0_rad = ch.lon0_deg * 0.017453292519943295;
    c.k0 = ch.k0; c.fe = ch.false_easting; c.fn = ch.false_northing;
  }
  const size_t bytes = sizeof(double) * (size_t)n;
  double* outs_h[6] = {ox, oy, lat, lon, ch.projected ? east : nullptr, ch.projected ? north : nullptr};
  int n_out = 0;
  for (double* p : outs_h) n_out += p != nullptr;
  DevBuf din(2 * bytes), dout(std::max<size_t>(n_out, 1) * bytes);
  hipStream_t s = ctx->stream;
  GTX_HIP(hipMemcpyAsync(din.p, x, bytes, hipMemcpyHostToDevice, s));
  GTX_HIP(hipMemcpyAsync(din.as<double>() + n, y, bytes, hipMemcpyHostToDevice, s));
  double* outs_d[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0, j = 0; k < 6; ++k)
    if (outs_h[k]) outs_d[k] = dout.as<double>() + (size_t)(j++) * n;
  hipLaunchKernelGGL(georef_points_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, c, din.as<double>(), din.as<double>() + n, n, outs_d[0],
                     outs_d[1], outs_d[2], outs_d[3], outs_d[4], outs_d[5]);
  GTX_HIP(hipGetLastError());
  for (int k = 0; k < 6; ++k)
    if (outs_h[k]) GTX_HIP(hipMemcpyAsync(outs_h[k], outs_d[k], bytes, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
}

}  // namespace gtx
