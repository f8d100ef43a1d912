// Global motion compensation, method 'ecc': see ecc.hpp. gfx950 only. Compiled with -ffp-contract=off: the float32 / float64
// operation sequences below are oracle/ecc_ref.py's (OpenCV's), one rounding per operation.
//
// Data in HBM: the template (first frame of the sequence) and a 32-deep ring of prepared frames, float32 [h/2][w/2] each
// (8.3 MB at 3840x2160), the current frame's two gradient images, 512 x 13 float64 partial sums. An iteration is four launches:
//   ecc_stats_kernel   every template pixel: warpAffine's fixed-point source coordinates, the bilinear sample of the frame, the
//                      nearest-neighbour validity mask; count / sum / sum of squares of the sample and of the template under the mask
//   ecc_stats_finish   the 512 partial sums in a fixed order -> the two means and standard deviations
//   ecc_accum_kernel   the samples again (frame, both gradients: 12 reads against 3 written + 3 read back), zero-mean values,
//                      the Euclidean Jacobian, and the 13 sums an iteration needs: Hessian (6), the two projections (3 + 3), the correlation
//   ecc_update_kernel  the 512 partial sums in a fixed order, then one thread: 3 x 3 inverse, rho, lambda, the parameter step, the new map
// Both passes are HBM-bound reads of the frame at the warped positions (16.6 MB + 50 MB per iteration at 4K: ~10 us + ~20 us at
// the rates this chip's gathers reach); 20-60 iterations per frame are typical for drone footage against its first frame.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>

#include "ecc.hpp"

namespace gtx {

namespace {

constexpr int kBlocks = 512, kThreads = 256, kRing = 32, kItersPerCheck = 6, kSums = 13;

struct EccState {
  float map[6];
  int iter, status, done, max_iters;
  int exact, pad_;                // 1: source positions in floating point (OpenCV >= 4.11), 0: warpAffine's fixed point (through 4.10)
  double eps, rho, last_rho;
  double n;                       // pixels under the mask
  double img_norm, tmp_norm;
  float img_mean, tmp_mean;       // the float32 the masked subtraction uses
};

__device__ __forceinline__ int refl101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// cvtColor(BGR2GRAY) -> GaussianBlur(3x3, sigma 1.5: the bit-exact kernel (79, 98, 79) / 256, BORDER_REFLECT_101, one rounding)
// -> resize to half size (2 x 2 mean, round half up), one output pixel per thread, as float32.
__global__ __launch_bounds__(256) void ecc_prepare_kernel(const uint8_t* __restrict__ bgr, int H, int W, float* __restrict__ out, int h2, int w2) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w2 || y >= h2) return;
  int g[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int yy = refl101(2 * y - 1 + r, H);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int xx = refl101(2 * x - 1 + c, W);
      const uint8_t* p = bgr + ((size_t)yy * W + xx) * 3;
      g[r][c] = (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14;
    }
  }
  int hz[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) hz[r][c] = 79 * g[r][c] + 98 * g[r][c + 1] + 79 * g[r][c + 2];
  int s = 0;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int v = 79 * hz[r][c] + 98 * hz[r + 1][c] + 79 * hz[r + 2][c];
      s += min((v + (1 << 15)) >> 16, 255);
    }
  out[(size_t)y * w2 + x] = (float)((s + 2) >> 2);
}

// filter2D with (-0.5, 0, 0.5) and its transpose, BORDER_REFLECT_101
__global__ __launch_bounds__(256) void ecc_gradient_kernel(const float* __restrict__ img, int h, int w, float* __restrict__ gx, float* __restrict__ gy) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const float c = img[(size_t)y * w + x];
  const float xm = img[(size_t)y * w + refl101(x - 1, w)], xp = img[(size_t)y * w + refl101(x + 1, w)];
  const float ym = img[(size_t)refl101(y - 1, h) * w + x], yp = img[(size_t)refl101(y + 1, h) * w + x];
  gx[(size_t)y * w + x] = (-0.5f * xm + 0.f * c) + 0.5f * xp;
  gy[(size_t)y * w + x] = (-0.5f * ym + 0.f * c) + 0.5f * yp;
}

// warpAffine(INTER_LINEAR | WARP_INVERSE_MAP) source position of template pixel (x, y), in the two forms OpenCV has had
// (oracle/ecc_ref.py): exact -- the position m0 x + m1 y + m2 in floating point, floor + fraction, p00 + a (p01 - p00) ... --
// or fixed point: integer pixel + 1/32 fractions, table weights. And the INTER_NEAREST pixel the mask uses (fixed point in both).
// m: the map as float64.
struct Src { int sx, sy, nx, ny; float ax, ay; };
__device__ __forceinline__ Src source_of(const double* m, int x, int y, bool exact) {
  const long long ad = __double2ll_rn(m[0] * (double)x * 1024.0), bd = __double2ll_rn(m[3] * (double)x * 1024.0);
  const long long X0 = __double2ll_rn((m[1] * (double)y + m[2]) * 1024.0), Y0 = __double2ll_rn((m[4] * (double)y + m[5]) * 1024.0);
  Src s;
  s.nx = (int)((X0 + 512 + ad) >> 10); s.ny = (int)((Y0 + 512 + bd) >> 10);
  if (exact) {
    const double fx = (m[0] * (double)x + m[1] * (double)y) + m[2], fy = (m[3] * (double)x + m[4] * (double)y) + m[5];
    const double ix = floor(fx), iy = floor(fy);
    s.sx = (int)ix; s.sy = (int)iy; s.ax = (float)(fx - ix); s.ay = (float)(fy - iy);
  } else {
    const long long xl = (X0 + 16 + ad) >> 5, yl = (Y0 + 16 + bd) >> 5;
    s.sx = (int)(xl >> 5); s.sy = (int)(yl >> 5);
    s.ax = (float)(int)(xl & 31) * (1.f / 32.f); s.ay = (float)(int)(yl & 31) * (1.f / 32.f);
  }
  return s;
}
__device__ __forceinline__ float fetch0(const float* __restrict__ a, int h, int w, int y, int x) {
  return (y >= 0 && y < h && x >= 0 && x < w) ? a[(size_t)y * w + x] : 0.f;
}
__device__ __forceinline__ float sample(const float* __restrict__ a, int h, int w, const Src& s, bool exact) {
  const float p00 = fetch0(a, h, w, s.sy, s.sx), p01 = fetch0(a, h, w, s.sy, s.sx + 1), p10 = fetch0(a, h, w, s.sy + 1, s.sx), p11 = fetch0(a, h, w, s.sy + 1, s.sx + 1);
  if (exact) {
    const float v0 = p00 + s.ax * (p01 - p00), v1 = p10 + s.ax * (p11 - p10);
    return v0 + s.ay * (v1 - v0);
  }
  const float w00 = (1.f - s.ay) * (1.f - s.ax), w01 = (1.f - s.ay) * s.ax, w10 = s.ay * (1.f - s.ax), w11 = s.ay * s.ax;
  return ((p00 * w00 + p01 * w01) + p10 * w10) + p11 * w11;
}

template <int N>
__device__ __forceinline__ void block_sums(double (&v)[N], double* __restrict__ out) {   // out[N] of this block; fixed order
  __shared__ double part[kThreads / 64][N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double t = v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double t = 0.0;
    for (int wv = 0; wv < kThreads / 64; ++wv) t += part[wv][threadIdx.x];
    out[threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(kThreads) void ecc_stats_kernel(const EccState* __restrict__ st, const float* __restrict__ img, const float* __restrict__ tmpl,
                                                             int h, int w, double* __restrict__ partial) {
  if (st->done) return;
  double m[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) m[i] = (double)st->map[i];
  double v[5] = {0, 0, 0, 0, 0};
  const bool exact = st->exact != 0;
  const int total = h * w;
  for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += kBlocks * kThreads) {
    const int y = idx / w, x = idx - y * w;
    const Src s = source_of(m, x, y, exact);
    if (!(s.ny >= 0 && s.ny < h && s.nx >= 0 && s.nx < w)) continue;
    const double a = (double)sample(img, h, w, s, exact), t = (double)tmpl[idx];
    v[0] += 1.0; v[1] += a; v[2] += a * a; v[3] += t; v[4] += t * t;
  }
  block_sums<5>(v, partial + (size_t)blockIdx.x * kSums);
}

template <int N>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partial, double (&out)[N]) {   // every thread gets the totals
  __shared__ double acc[kThreads][N];
  for (int i = 0; i < N; ++i) {
    double t = 0.0;
    for (int b = threadIdx.x; b < kBlocks; b += kThreads) t += partial[(size_t)b * kSums + i];
    acc[threadIdx.x][i] = t;
  }
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int i = 0; i < N; ++i) acc[threadIdx.x][i] += acc[threadIdx.x + o][i];
    __syncthreads();
  }
  for (int i = 0; i < N; ++i) out[i] = acc[0][i];
}

__global__ __launch_bounds__(kThreads) void ecc_stats_finish_kernel(EccState* __restrict__ st, const double* __restrict__ partial) {
  if (st->done) return;
  double s[5];
  reduce_partials<5>(partial, s);
  if (threadIdx.x != 0) return;
  const double n = s[0];
  double im = 0.0, is = 0.0, tm = 0.0, ts = 0.0;
  if (n > 0.0) {                                   // meanStdDev: float64 sums, population variance
    im = s[1] / n; is = sqrt(fmax(s[2] / n - im * im, 0.0));
    tm = s[3] / n; ts = sqrt(fmax(s[4] / n - tm * tm, 0.0));
  }
  st->n = n;
  st->img_mean = (float)im; st->tmp_mean = (float)tm;
  st->img_norm = sqrt(n * is * is);
  st->tmp_norm = sqrt(n * ts * ts);
}

__global__ __launch_bounds__(kThreads) void ecc_accum_kernel(const EccState* __restrict__ st, const float* __restrict__ img, const float* __restrict__ gx,
                                                             const float* __restrict__ gy, const float* __restrict__ tmpl, int h, int w,
                                                             double* __restrict__ partial) {
  if (st->done) return;
  double m[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) m[i] = (double)st->map[i];
  const float h0 = st->map[0], h1 = st->map[3], img_mean = st->img_mean, tmp_mean = st->tmp_mean;
  const bool exact = st->exact != 0;
  double v[kSums];
#pragma unroll
  for (int i = 0; i < kSums; ++i) v[i] = 0.0;
  const int total = h * w;
  for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += kBlocks * kThreads) {
    const int y = idx / w, x = idx - y * w;
    const Src s = source_of(m, x, y, exact);
    const bool mask = s.ny >= 0 && s.ny < h && s.nx >= 0 && s.nx < w;
    float iw = sample(img, h, w, s, exact);
    const float gxw = sample(gx, h, w, s, exact), gyw = sample(gy, h, w, s, exact);
    float tz = 0.f;
    if (mask) { iw = iw - img_mean; tz = tmpl[idx] - tmp_mean; }
    const float X = (float)x, Y = (float)y;
    const float hat_x = -(X * h1) - (Y * h0), hat_y = (X * h0) - (Y * h1);
    const double j0 = (double)((gxw * hat_x) + (gyw * hat_y)), j1 = (double)gxw, j2 = (double)gyw;
    const double iwd = (double)iw, tzd = (double)tz;
    v[0] += j0 * j0; v[1] += j0 * j1; v[2] += j0 * j2; v[3] += j1 * j1; v[4] += j1 * j2; v[5] += j2 * j2;
    v[6] += j0 * iwd; v[7] += j1 * iwd; v[8] += j2 * iwd;
    v[9] += j0 * tzd; v[10] += j1 * tzd; v[11] += j2 * tzd;
    v[12] += tzd * iwd;
  }
  block_sums<kSums>(v, partial + (size_t)blockIdx.x * kSums);
}

__global__ __launch_bounds__(kThreads) void ecc_update_kernel(EccState* __restrict__ st, const double* __restrict__ partial) {
  if (st->done) return;
  double s[kSums];
  reduce_partials<kSums>(partial, s);
  if (threadIdx.x != 0) return;
  st->iter += 1;
  // the Hessian and the projections are float32 matrices upstream (the dot products that fill them are float64)
  const float hs[6] = {(float)s[0], (float)s[1], (float)s[2], (float)s[3], (float)s[4], (float)s[5]};
  const double a00 = hs[0], a01 = hs[1], a02 = hs[2], a11 = hs[3], a12 = hs[4], a22 = hs[5];
  const double a10 = a01, a20 = a02, a21 = a12;
  float inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // cv::invert(DECOMP_LU) of a 3 x 3: cofactors in float64; singular -> zeros
  double d = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
  if (d != 0.0) {
    d = 1.0 / d;
    inv[0] = (float)((a11 * a22 - a12 * a21) * d); inv[1] = (float)((a02 * a21 - a01 * a22) * d); inv[2] = (float)((a01 * a12 - a02 * a11) * d);
    inv[3] = (float)((a12 * a20 - a10 * a22) * d); inv[4] = (float)((a00 * a22 - a02 * a20) * d); inv[5] = (float)((a02 * a10 - a00 * a12) * d);
    inv[6] = (float)((a10 * a21 - a11 * a20) * d); inv[7] = (float)((a01 * a20 - a00 * a21) * d); inv[8] = (float)((a00 * a11 - a01 * a10) * d);
  }
  const double correlation = s[12];
  const double img_norm = st->img_norm, tmp_norm = st->tmp_norm;
  st->last_rho = st->rho;
  const double den = img_norm * tmp_norm;
  const double rho = den != 0.0 ? correlation / den : nan("");
  st->rho = rho;
  if (isnan(rho)) { st->status = 1; st->done = 1; return; }
  const float ip[3] = {(float)s[6], (float)s[7], (float)s[8]}, tp[3] = {(float)s[9], (float)s[10], (float)s[11]};
  float iph[3];
  for (int i = 0; i < 3; ++i) iph[i] = (float)(((double)inv[3 * i] * ip[0] + (double)inv[3 * i + 1] * ip[1]) + (double)inv[3 * i + 2] * ip[2]);
  const double lambda_n = img_norm * img_norm - (((double)ip[0] * iph[0] + (double)ip[1] * iph[1]) + (double)ip[2] * iph[2]);
  const double lambda_d = correlation - (((double)tp[0] * iph[0] + (double)tp[1] * iph[1]) + (double)tp[2] * iph[2]);
  if (lambda_d <= 0.0) { st->rho = -1.0; st->status = 2; st->done = 1; return; }
  const double lambda = lambda_n / lambda_d;
  float ep[3], dp[3];
  for (int i = 0; i < 3; ++i) ep[i] = (float)(lambda * (double)tp[i] - (double)ip[i]);
  for (int i = 0; i < 3; ++i) dp[i] = (float)(((double)inv[3 * i] * ep[0] + (double)inv[3 * i + 1] * ep[1]) + (double)inv[3 * i + 2] * ep[2]);
  const double theta = asin((double)st->map[3]) + (double)dp[0];
  st->map[2] = (float)((double)st->map[2] + (double)dp[1]);
  st->map[5] = (float)((double)st->map[5] + (double)dp[2]);
  st->map[0] = st->map[4] = (float)cos(theta);
  st->map[3] = (float)sin(theta);
  st->map[1] = -st->map[3];
  if (st->iter >= st->max_iters || fabs(st->rho - st->last_rho) < st->eps) st->done = 1;
}

}  // namespace

struct Ecc::Impl {
  int device;
  hipStream_t stream;
  int H, W, h2, w2, max_iters;
  double eps;
  DevBuf ring, tmpl, gx, gy, partial, state, stage;
  EccState* h_state = nullptr;                     // pinned
  hipEvent_t ready[kRing];
  hipEvent_t done_ev;
  std::mutex mu;
  long submitted = 0, collected = 0;
  bool have_template = false, replace_template = false, exact = true;
  int last_slot = -1;

  float* slot(long i) const { return ring.as<float>() + (size_t)(i % kRing) * h2 * w2; }
};

Ecc::Ecc(int device, hipStream_t stream, int frame_h, int frame_w, int max_iters, double eps) : impl_(new Impl) {
  Impl& S = *impl_;
  GTX_CHECK(frame_h >= 8 && frame_w >= 8 && max_iters >= 1 && eps > 0.0, "ecc: frame %dx%d, %d iterations, eps %g", frame_w, frame_h, max_iters, eps);
  S.device = device; S.stream = stream; S.H = frame_h; S.W = frame_w; S.h2 = frame_h / 2; S.w2 = frame_w / 2; S.max_iters = max_iters; S.eps = eps;
  GTX_HIP(hipSetDevice(device));
  const size_t px = (size_t)S.h2 * S.w2;
  S.ring.alloc(px * 4 * kRing); S.tmpl.alloc(px * 4); S.gx.alloc(px * 4); S.gy.alloc(px * 4);
  S.partial.alloc((size_t)kBlocks * kSums * 8); S.state.alloc(sizeof(EccState));
  GTX_HIP(hipHostMalloc((void**)&S.h_state, sizeof(EccState)));
  for (auto& e : S.ready) GTX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  GTX_HIP(hipEventCreateWithFlags(&S.done_ev, wait_event_flags(false)));
}

Ecc::~Ecc() {
  Impl& S = *impl_;
  (void)hipSetDevice(S.device);
  (void)hipStreamSynchronize(S.stream);
  for (auto& e : S.ready) (void)hipEventDestroy(e);
  (void)hipEventDestroy(S.done_ev);
  if (S.h_state) (void)hipHostFree(S.h_state);
}

void Ecc::reset() {
  Impl& S = *impl_;
  std::lock_guard<std::mutex> lk(S.mu);
  S.collected = S.submitted;                       // frames submitted and never collected are dropped
  S.have_template = false;
  S.last_slot = -1;
}

void Ecc::set_replace_template(bool on) { impl_->replace_template = on; }
void Ecc::set_exact_positions(bool on) { impl_->exact = on; }

int Ecc::pending() const {
  Impl& S = *impl_;
  std::lock_guard<std::mutex> lk(S.mu);
  return (int)(S.submitted - S.collected);
}

void Ecc::submit_frame_dev(const void* frame, int h, int w, hipStream_t producer) {
  Impl& S = *impl_;
  GTX_CHECK(frame && h == S.H && w == S.W, "ecc: frame is %dx%d, the object was made for %dx%d", w, h, S.W, S.H);
  GTX_HIP(hipSetDevice(S.device));
  long i;
  {
    std::lock_guard<std::mutex> lk(S.mu);
    GTX_CHECK(S.submitted - S.collected < kRing, "ecc: %d frames are waiting to be collected (the ring holds %d)", (int)(S.submitted - S.collected), kRing);
    i = S.submitted;
  }
  hipLaunchKernelGGL(ecc_prepare_kernel, dim3(cdiv(S.w2, 64), cdiv(S.h2, 4)), dim3(256), 0, producer, static_cast<const uint8_t*>(frame), S.H, S.W, S.slot(i), S.h2, S.w2);
  GTX_HIP(hipGetLastError());
  GTX_HIP(hipEventRecord(S.ready[i % kRing], producer));
  std::lock_guard<std::mutex> lk(S.mu);
  S.submitted = i + 1;
}

void Ecc::submit_frame(const uint8_t* frame, int h, int w) {
  Impl& S = *impl_;
  GTX_CHECK(frame && h == S.H && w == S.W, "ecc: frame is %dx%d, the object was made for %dx%d", w, h, S.W, S.H);
  GTX_HIP(hipSetDevice(S.device));
  const size_t bytes = (size_t)h * w * 3;
  if (S.stage.bytes < bytes) { GTX_HIP(hipStreamSynchronize(S.stream)); S.stage.alloc(bytes); }
  GTX_HIP(hipStreamSynchronize(S.stream));         // the staging buffer's previous frame has been prepared
  GTX_HIP(hipMemcpyAsync(S.stage.p, frame, bytes, hipMemcpyHostToDevice, S.stream));
  submit_frame_dev(S.stage.p, h, w, S.stream);
}

void Ecc::collect(double A[6], int info[2], double* rho) {
  Impl& S = *impl_;
  GTX_HIP(hipSetDevice(S.device));
  long i;
  {
    std::lock_guard<std::mutex> lk(S.mu);
    GTX_CHECK(S.collected < S.submitted, "ecc: collect without a submitted frame");
    i = S.collected;
  }
  const double ident[6] = {1, 0, 0, 0, 1, 0};
  for (int k = 0; k < 6; ++k) A[k] = ident[k];
  if (info) { info[0] = 0; info[1] = 0; }
  if (rho) *rho = 0.0;
  const size_t px = (size_t)S.h2 * S.w2;
  GTX_HIP(hipStreamWaitEvent(S.stream, S.ready[i % kRing], 0));
  if (!S.have_template) {
    GTX_HIP(hipMemcpyAsync(S.tmpl.p, S.slot(i), px * 4, hipMemcpyDeviceToDevice, S.stream));
    GTX_HIP(hipStreamSynchronize(S.stream));
    S.have_template = true;
  } else {
    const float* img = S.slot(i);
    hipLaunchKernelGGL(ecc_gradient_kernel, dim3(cdiv(S.w2, 64), cdiv(S.h2, 4)), dim3(256), 0, S.stream, img, S.h2, S.w2, S.gx.as<float>(), S.gy.as<float>());
    EccState init{};
    init.map[0] = init.map[4] = 1.f;
    init.max_iters = S.max_iters; init.eps = S.eps; init.rho = -1.0; init.last_rho = -S.eps; init.exact = S.exact ? 1 : 0;
    *S.h_state = init;
    GTX_HIP(hipMemcpyAsync(S.state.p, S.h_state, sizeof(EccState), hipMemcpyHostToDevice, S.stream));
    EccState* st = S.state.as<EccState>();
    double* part = S.partial.as<double>();
    for (int launched = 0; launched < S.max_iters;) {
      for (int k = 0; k < kItersPerCheck && launched < S.max_iters; ++k, ++launched) {
        hipLaunchKernelGGL(ecc_stats_kernel, dim3(kBlocks), dim3(kThreads), 0, S.stream, st, img, S.tmpl.as<float>(), S.h2, S.w2, part);
        hipLaunchKernelGGL(ecc_stats_finish_kernel, dim3(1), dim3(kThreads), 0, S.stream, st, part);
        hipLaunchKernelGGL(ecc_accum_kernel, dim3(kBlocks), dim3(kThreads), 0, S.stream, st, img, S.gx.as<float>(), S.gy.as<float>(), S.tmpl.as<float>(), S.h2, S.w2, part);
        hipLaunchKernelGGL(ecc_update_kernel, dim3(1), dim3(kThreads), 0, S.stream, st, part);
      }
      GTX_HIP(hipGetLastError());
      GTX_HIP(hipMemcpyAsync(S.h_state, S.state.p, sizeof(EccState), hipMemcpyDeviceToHost, S.stream));
      GTX_HIP(hipEventRecord(S.done_ev, S.stream));
      GTX_HIP(hipEventSynchronize(S.done_ev));
      if (S.h_state->done) break;
    }
    for (int k = 0; k < 6; ++k) A[k] = (double)S.h_state->map[k];
    if (info) { info[0] = S.h_state->iter; info[1] = S.h_state->status; }
    if (rho) *rho = S.h_state->rho;
    if (S.replace_template) {
      GTX_HIP(hipMemcpyAsync(S.tmpl.p, img, px * 4, hipMemcpyDeviceToDevice, S.stream));
      GTX_HIP(hipStreamSynchronize(S.stream));     // the slot may be reused by a later submission
    }
  }
  std::lock_guard<std::mutex> lk(S.mu);
  S.last_slot = (int)(i % kRing);
  S.collected = i + 1;
}

void Ecc::debug_image(int which, float* out) const {
  Impl& S = *impl_;
  GTX_HIP(hipSetDevice(S.device));
  GTX_CHECK(which == 1 ? S.have_template : S.last_slot >= 0, "ecc: no image of kind %d yet", which);
  const float* src = which == 1 ? S.tmpl.as<float>() : S.ring.as<float>() + (size_t)S.last_slot * S.h2 * S.w2;
  GTX_HIP(hipStreamSynchronize(S.stream));
  GTX_HIP(hipMemcpy(out, src, (size_t)S.h2 * S.w2 * 4, hipMemcpyDeviceToHost));
}

}  // namespace gtx
