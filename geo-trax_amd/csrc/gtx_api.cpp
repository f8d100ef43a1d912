// C ABI of libgtx.so (see include/gtx.h). Everything here is a thin try/catch shim that turns
// gtx::Error into a status code + thread-local message.
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>

#include "../../include/gtx.h"
#include "api_guard.hpp"
#include "common.hpp"
#include "conv_igemm.hpp"
#include "det_kernels.hpp"
#include "detector.hpp"
#include "rtdetr.hpp"
#include "geometry.hpp"
#include "ecc.hpp"
#include "gmc.hpp"
#include "match_l2.hpp"
#include "register.hpp"
#include "sift.hpp"
#include "split_format.hpp"
#include "stabilizer.hpp"
#include "tracker.hpp"

namespace {
using gtx::g_last_error;
using gtx::guarded;

void need(const void* p, const char* what) {
  if (!p) gtx::fail(GTX_ERR_INVALID, "%s is NULL", what);
}
}  // namespace

extern "C" {

int gtx_abi_version(void) { return GTX_ABI_VERSION; }
const char* gtx_last_error(void) { return g_last_error.c_str(); }

int gtx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int gtx_ctx_create(int device, gtx_ctx** out) { return gtx_ctx_create_prio(device, 0, out); }

int gtx_ctx_create_prio(int device, int high_priority, gtx_ctx** out) {
  return guarded([&] {
    need(out, "out");
    int n = 0;
    GTX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) gtx::fail(GTX_ERR_INVALID, "device %d not in [0,%d)", device, n);
    GTX_HIP(hipSetDevice(device));
    std::unique_ptr<gtx_ctx> c(new gtx_ctx);
    c->device = device;
    GTX_HIP(hipGetDeviceProperties(&c->prop, device));
    if (std::string(c->prop.gcnArchName).find("gfx950") == std::string::npos)
      gtx::fail(GTX_ERR_UNSUPPORTED, "libgtx is built for gfx950 only; device %d is %s", device, c->prop.gcnArchName);
    if (high_priority != 0) {        // > 0: the device's highest stream priority, < 0: its lowest
      int least = 0, greatest = 0;   // numerically lower = higher priority
      GTX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      GTX_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, high_priority > 0 ? greatest : least));
    } else {
      GTX_HIP(hipStreamCreate(&c->stream));
    }
    *out = c.release();
  });
}

void gtx_ctx_destroy(gtx_ctx* ctx) { delete ctx; }

namespace {
// one wave that does nothing for `ticks` of the 100 MHz constant clock
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t = t0;
  while (t - t0 < ticks) {
    __builtin_amdgcn_s_sleep(16);
    t = __builtin_amdgcn_s_memrealtime();
  }
  if (sink && threadIdx.x == 0 && ticks == ~0ull) *sink = t;
}
}  // namespace

int gtx_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
  return guarded([&] {
    need(free_bytes, "free_bytes"); need(total_bytes, "total_bytes");
    GTX_HIP(hipSetDevice(device));
    GTX_HIP(hipMemGetInfo(free_bytes, total_bytes));
  });
}

int gtx_streams_overlap(gtx_ctx* a, gtx_ctx* b, float spin_us, float* ms_single, float* ms_pair) {
  return guarded([&] {
    need(a, "a"); need(b, "b"); need(ms_single, "ms_single"); need(ms_pair, "ms_pair");
    if (a->device != b->device) gtx::fail(GTX_ERR_INVALID, "the two contexts are on devices %d and %d", a->device, b->device);
    GTX_HIP(hipSetDevice(a->device));
    const unsigned long long ticks = (unsigned long long)(std::max(spin_us, 1.f) * 100.f);
    auto timed = [&](bool both) {
      GTX_HIP(hipStreamSynchronize(a->stream));
      GTX_HIP(hipStreamSynchronize(b->stream));
      const auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a->stream, ticks, (unsigned long long*)nullptr);
      if (both) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b->stream, ticks, (unsigned long long*)nullptr);
      GTX_HIP(hipGetLastError());
      GTX_HIP(hipStreamSynchronize(a->stream));
      GTX_HIP(hipStreamSynchronize(b->stream));
      return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    (void)timed(true);                       // the code object is loaded, both streams have run something
    float one = 1e30f, two = 1e30f;
    for (int i = 0; i < 3; ++i) {            // host-timed: the best of three is the one without a scheduling hiccup
      one = std::min(one, timed(false));
      two = std::min(two, timed(true));
    }
    *ms_single = one;
    *ms_pair = two;
  });
}

int gtx_device_open_null_stream(int device) {
  return guarded([&] {
    GTX_HIP(hipSetDevice(device));
    void* p = nullptr;
    GTX_HIP(hipMalloc(&p, 256));
    hipError_t e = hipMemset(p, 0, 4);            // synchronous: runs on (and thereby creates) the null stream
    (void)hipFree(p);
    GTX_HIP(e);
  });
}

int gtx_ctx_synchronize(gtx_ctx* ctx) {
  return guarded([&] {
    need(ctx, "ctx");
    GTX_HIP(hipSetDevice(ctx->device));
    GTX_HIP(hipStreamSynchronize(ctx->stream));
  });
}

int gtx_dev_alloc(gtx_ctx* ctx, size_t bytes, void** dptr) {
  return guarded([&] {
    need(ctx, "ctx");
    need(dptr, "dptr");
    GTX_HIP(hipSetDevice(ctx->device));
    GTX_HIP(hipMalloc(dptr, bytes ? bytes : 256));
  });
}
int gtx_dev_free(gtx_ctx* ctx, void* dptr) {
  return guarded([&] {
    need(ctx, "ctx");
    GTX_HIP(hipSetDevice(ctx->device));
    if (dptr) GTX_HIP(hipFree(dptr));
  });
}
int gtx_dev_upload(gtx_ctx* ctx, void* dptr, const void* host, size_t bytes) {
  return guarded([&] {
    need(ctx, "ctx"); need(dptr, "dptr"); need(host, "host");
    GTX_HIP(hipSetDevice(ctx->device));
    GTX_HIP(hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    GTX_HIP(hipStreamSynchronize(ctx->stream));
  });
}
int gtx_dev_download(gtx_ctx* ctx, void* host, const void* dptr, size_t bytes) {
  return guarded([&] {
    need(ctx, "ctx"); need(dptr, "dptr"); need(host, "host");
    GTX_HIP(hipSetDevice(ctx->device));
    GTX_HIP(hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GTX_HIP(hipStreamSynchronize(ctx->stream));
  });
}

/* ------------------------------------------------------------------ operator level */

namespace {
struct ConvOpState {
  gtx::DevBuf x, w, b, r, y;
  gtx::ConvGroup g{};
  gtx::ConvConfig cfg{};
  int ho = 0, wo = 0;
};

void conv_setup(gtx_ctx* ctx, const gtx_conv_desc* d, const void* x, const float* w, const float* bias,
                const void* residual, const void* y_init, ConvOpState& st) {
  using namespace gtx;
  need(ctx, "ctx"); need(d, "desc");
  GTX_HIP(hipSetDevice(ctx->device));
  if (d->dtype != GTX_F16 && d->dtype != GTX_F32 && d->dtype != GTX_F32S) fail(GTX_ERR_INVALID, "bad dtype %d", d->dtype);
  const size_t es = dtype_size(d->dtype);
  const int pad = d->ksize / 2;
  st.ho = (d->h + 2 * pad - d->ksize) / d->stride + 1;
  st.wo = (d->w + 2 * pad - d->ksize) / d->stride + 1;
  st.cfg = conv_pick_config(d->dtype, d->ksize, d->stride, d->cin, d->cout);
  const bool pairs = d->dtype == GTX_F32S;       // host arrays are plain fp32; the device buffers hold the pair format (split_format.hpp)
  const int vn = pairs ? 8 : 16 / (int)es;
  GTX_CHECK(d->in_cstride % vn == 0 && d->in_coff % vn == 0 && d->out_cstride % (pairs ? 8 : 4) == 0 && d->out_coff % (pairs ? 8 : 4) == 0 &&
                (!pairs || d->cout % 8 == 0),
            "conv: channel strides/offsets must keep 16-byte (input) / 4-element (output) alignment, whole 8-channel groups for the split-f16x3 path");
  GTX_CHECK(d->in_coff + d->cin <= d->in_cstride && d->out_coff + d->cout <= d->out_cstride, "conv: slice outside buffer");
  const size_t xin = (size_t)d->n * d->h * d->w * d->in_cstride * es;
  const size_t yout = (size_t)d->n * st.ho * st.wo * d->out_cstride * es;
  st.x.alloc(xin);
  st.y.alloc(yout);
  // timing calls (no data handed in) run on pseudo-random activations and weights: zeros would flatter the matrix pipe
  // (no operand toggling, no power throttling) -- the layer sweeps of rounds 1 and 2 up to this change were taken on zeros
  unsigned long long lcg = 0x2545F4914F6CDD1Dull;
  const bool zeros = std::getenv("GTX_TIME_ZEROS") != nullptr;      // the old behaviour, to show the difference
  auto uni = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return zeros ? 0.f : (float)((lcg >> 40) * (1.0 / 8388608.0) - 1.0); };
  auto upload = [&](void* dst, const void* src, size_t bytes) {           // plain fp32 host array -> pair format on the device
    if (!pairs) { GTX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return; }
    std::vector<uint8_t> tmp(bytes);
    f32_to_pairs(static_cast<const float*>(src), tmp.data(), bytes / 4);
    GTX_HIP(hipMemcpy(dst, tmp.data(), bytes, hipMemcpyHostToDevice));
  };
  if (x) {
    upload(st.x.p, x, xin);
  } else if (es == 2) {
    std::vector<_Float16> hx(xin / 2);
    for (auto& v : hx) v = (_Float16)uni();
    GTX_HIP(hipMemcpy(st.x.p, hx.data(), xin, hipMemcpyHostToDevice));
  } else {
    std::vector<float> hx(xin / 4);
    for (auto& v : hx) v = uni();
    upload(st.x.p, hx.data(), xin);
  }
  if (y_init) upload(st.y.p, y_init, yout);
  std::vector<uint8_t> packed;
  float acc_scale = 1.f;
  if (w) {
    packed = pack_conv_weights(w, d->cout, d->cin, st.cfg, &acc_scale);
  } else {
    std::vector<float> hw((size_t)d->cout * d->cin * d->ksize * d->ksize);
    const float sc = 1.f / std::sqrt((float)(d->cin * d->ksize * d->ksize));
    for (auto& v : hw) v = uni() * sc;
    packed = pack_conv_weights(hw.data(), d->cout, d->cin, st.cfg, &acc_scale);
  }
  st.w.alloc(packed.size());
  GTX_HIP(hipMemcpy(st.w.p, packed.data(), packed.size(), hipMemcpyHostToDevice));
  if (bias) {
    const size_t padded = (size_t)(d->cout + 63) / 64 * 64 * sizeof(float);      // whole cout tiles: the kernels load a tile's bias unconditionally
    st.b.alloc(padded);
    GTX_HIP(hipMemset(st.b.p, 0, padded));
    GTX_HIP(hipMemcpy(st.b.p, bias, d->cout * sizeof(float), hipMemcpyHostToDevice));
  }
  if (d->has_residual) {
    const size_t rb = (size_t)d->n * st.ho * st.wo * d->cout * es;
    st.r.alloc(rb);
    if (residual) upload(st.r.p, residual, rb);
    else GTX_HIP(hipMemset(st.r.p, 0, rb));
  }
  ConvProblem& p = st.g.p[0];
  p.in = st.x.p; p.out = st.y.p; p.wpack = st.w.p;
  p.bias = bias ? st.b.as<float>() : nullptr;
  p.res = d->has_residual ? st.r.p : nullptr;
  p.N = d->n; p.H = d->h; p.W = d->w; p.Ho = st.ho; p.Wo = st.wo; p.Cin = d->cin; p.Cout = d->cout;
  p.in_cstride = d->in_cstride; p.in_coff = d->in_coff;
  p.out_cstride = d->out_cstride; p.out_coff = d->out_coff;
  p.res_cstride = d->cout; p.res_coff = 0;
  p.act = d->act;
  p.acc_scale = acc_scale;
  p.in2 = nullptr; p.in2_cstride = p.in2_coff = p.c_split = 0;
  p.out_plain = 0; p.sat_flag = nullptr;
  p.post_w = nullptr; p.post_bias = nullptr; p.post_scale = 1.f; p.post_act = 0;
  st.g.count = 1;
  conv_group_finalize(st.g, st.cfg);
}
}  // namespace

int gtx_op_conv2d(gtx_ctx* ctx, const gtx_conv_desc* d, const void* x, const float* w_ohwi, const float* bias,
                  const void* residual, void* y) {
  return guarded([&] {
    need(x, "x"); need(w_ohwi, "w"); need(y, "y");
    if (d && d->has_residual) need(residual, "residual");
    ConvOpState st;
    conv_setup(ctx, d, x, w_ohwi, bias, residual, y, st);
    gtx::conv_launch(st.g, st.cfg, ctx->stream);
    GTX_HIP(hipStreamSynchronize(ctx->stream));
    const size_t yb = (size_t)d->n * st.ho * st.wo * d->out_cstride * gtx::dtype_size(d->dtype);
    if (d->dtype == GTX_F32S) {                       // pair format -> the caller's plain fp32 array
      std::vector<uint8_t> tmp(yb);
      GTX_HIP(hipMemcpy(tmp.data(), st.y.p, yb, hipMemcpyDeviceToHost));
      gtx::pairs_to_f32(tmp.data(), static_cast<float*>(y), yb / 4);
    } else {
      GTX_HIP(hipMemcpy(y, st.y.p, yb, hipMemcpyDeviceToHost));
    }
  });
}

int gtx_op_conv_xcd_ranges(int n_members, const int* blocks, const int* cin, int xcd_begin[9], int* grid_blocks) {
  return guarded([&] {
    need(blocks, "blocks"); need(cin, "cin"); need(xcd_begin, "xcd_begin");
    if (n_members < 1 || n_members > gtx::kMaxGroup) gtx::fail(GTX_ERR_INVALID, "1..%d members", gtx::kMaxGroup);
    gtx::ConvGroup g{};
    gtx::ConvConfig c{};
    c.bn = 64; c.th = 8; c.tw = 16;                       // one workgroup per (8 x 16 pixel tile, 64-cout tile): blocks[i] = tiles_y
    g.count = n_members;
    for (int i = 0; i < n_members; ++i) {
      if (blocks[i] < 1 || cin[i] < 1) gtx::fail(GTX_ERR_INVALID, "member %d: blocks and cin must be positive", i);
      gtx::ConvProblem& p = g.p[i];
      p.N = 1; p.Wo = 16; p.Ho = 8 * blocks[i]; p.Cout = 64; p.Cin = cin[i];
    }
    gtx::conv_group_finalize(g, c);
    for (int k = 0; k < 9; ++k) xcd_begin[k] = g.xcd_begin[k];
    if (grid_blocks) *grid_blocks = g.grid_blocks;
  });
}

int gtx_op_conv2d_time(gtx_ctx* ctx, const gtx_conv_desc* d, int iters, float* ms_per_launch, double* flops) {
  return guarded([&] {
    need(ms_per_launch, "ms_per_launch");
    if (iters < 1) gtx::fail(GTX_ERR_INVALID, "iters must be >= 1");
    ConvOpState st;
    conv_setup(ctx, d, nullptr, nullptr, nullptr, nullptr, nullptr, st);
    hipEvent_t e0, e1;
    GTX_HIP(hipEventCreate(&e0));
    GTX_HIP(hipEventCreate(&e1));
    auto once = [&] {
      gtx::conv_launch(st.g, st.cfg, ctx->stream);
      };
    for (int i = 0; i < 3; ++i) once();
    GTX_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters; ++i) once();
    GTX_HIP(hipEventRecord(e1, ctx->stream));
    GTX_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    GTX_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    if (flops) *flops = gtx::conv_flops(st.g.p[0], d->ksize);
  });
}

int gtx_op_sppf_pool(gtx_ctx* ctx, int dtype, int n, int h, int w, int c, void* x_inout) {
  return guarded([&] {
    need(ctx, "ctx"); need(x_inout, "x");
    GTX_HIP(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n * h * w * 4 * c * gtx::dtype_size(dtype);
    gtx::DevBuf d(bytes);
    std::vector<uint8_t> tmp;
    if (dtype == GTX_F32S) {                          // plain fp32 host array <-> pair format on the device
      tmp.resize(bytes);
      gtx::f32_to_pairs(static_cast<const float*>(x_inout), tmp.data(), bytes / 4);
    }
    GTX_HIP(hipMemcpy(d.p, dtype == GTX_F32S ? tmp.data() : x_inout, bytes, hipMemcpyHostToDevice));
    gtx::launch_sppf_pool(dtype, d.p, n, h, w, c, ctx->stream);
    GTX_HIP(hipStreamSynchronize(ctx->stream));
    GTX_HIP(hipMemcpy(dtype == GTX_F32S ? tmp.data() : x_inout, d.p, bytes, hipMemcpyDeviceToHost));
    if (dtype == GTX_F32S) gtx::pairs_to_f32(tmp.data(), static_cast<float*>(x_inout), bytes / 4);
  });
}

int gtx_op_upsample2x(gtx_ctx* ctx, int dtype, int n, int h, int w, int c, const void* x, int in_cstride,
                      int in_coff, void* y, int out_cstride, int out_coff) {
  return guarded([&] {
    need(ctx, "ctx"); need(x, "x"); need(y, "y");
    GTX_HIP(hipSetDevice(ctx->device));
    const size_t es = gtx::dtype_size(dtype);
    const size_t xb = (size_t)n * h * w * in_cstride * es, yb = (size_t)n * 4 * h * w * out_cstride * es;
    gtx::DevBuf dx(xb), dy(yb);
    GTX_HIP(hipMemcpy(dx.p, x, xb, hipMemcpyHostToDevice));
    GTX_HIP(hipMemcpy(dy.p, y, yb, hipMemcpyHostToDevice));
    gtx::launch_upsample2x(dtype, dx.p, n, h, w, c, in_cstride, in_coff, dy.p, out_cstride, out_coff, ctx->stream);
    GTX_HIP(hipStreamSynchronize(ctx->stream));
    GTX_HIP(hipMemcpy(y, dy.p, yb, hipMemcpyDeviceToHost));
  });
}

struct gtx_gmc {
  gtx_ctx* ctx;
  std::unique_ptr<gtx::Gmc> impl;
};

int gtx_gmc_create(gtx_ctx* ctx, int frame_h, int frame_w, int seed, gtx_gmc** out) {
  return guarded([&] {
    need(ctx, "ctx"); need(out, "out");
    GTX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<gtx_gmc> g(new gtx_gmc);
    g->ctx = ctx;
    g->impl.reset(new gtx::Gmc(ctx->device, ctx->stream, frame_h / 2, frame_w / 2, seed));
    *out = g.release();
  });
}
void gtx_gmc_destroy(gtx_gmc* g) { delete g; }
int gtx_gmc_reset(gtx_gmc* g) {
  return guarded([&] { need(g, "gmc"); g->impl->reset(); });
}
int gtx_gmc_apply(gtx_gmc* g, const uint8_t* frame_bgr, int h, int w, double A[6], int* valid, int stats[3]) {
  return guarded([&] {
    need(g, "gmc"); need(frame_bgr, "frame"); need(A, "A");
    g->impl->submit_frame(frame_bgr, h, w);
    g->impl->collect(A, valid, stats);
  });
}
int gtx_gmc_submit_gray_dev(gtx_gmc* g, const void* gray_dptr, int gh, int gw) {
  return guarded([&] { need(g, "gmc"); need(gray_dptr, "gray"); g->impl->submit_gray_dev(gray_dptr, gh, gw); });
}
int gtx_gmc_restart(gtx_gmc* g) {
  return guarded([&] { need(g, "gmc"); g->impl->restart(); });
}
int gtx_gmc_submit_frame_dev(gtx_gmc* g, const void* frame_bgr_dptr, int h, int w, int restart) {
  return guarded([&] { need(g, "gmc"); need(frame_bgr_dptr, "frame"); g->impl->submit_frame_dev(frame_bgr_dptr, h, w, restart != 0); });
}
int gtx_gmc_collect(gtx_gmc* g, double A[6], int* valid, int stats[3]) {
  return guarded([&] { need(g, "gmc"); need(A, "A"); g->impl->collect(A, valid, stats); });
}
int gtx_gmc_points(gtx_gmc* g, int which, int cap, int* n, float* xy, int* status) {
  return guarded([&] { need(g, "gmc"); need(n, "n"); g->impl->debug_points(which, cap, n, xy, status); });
}

struct gtx_ecc {
  gtx_ctx* ctx;
  std::unique_ptr<gtx::Ecc> impl;
};

int gtx_ecc_create(gtx_ctx* ctx, int frame_h, int frame_w, int max_iters, double eps, gtx_ecc** out) {
  return guarded([&] {
    need(ctx, "ctx"); need(out, "out");
    GTX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<gtx_ecc> e(new gtx_ecc);
    e->ctx = ctx;
    e->impl.reset(new gtx::Ecc(ctx->device, ctx->stream, frame_h, frame_w, max_iters, eps));
    *out = e.release();
  });
}
void gtx_ecc_destroy(gtx_ecc* e) { delete e; }
int gtx_ecc_reset(gtx_ecc* e) {
  return guarded([&] { need(e, "ecc"); e->impl->reset(); });
}
int gtx_ecc_replace_template(gtx_ecc* e, int replace) {
  return guarded([&] { need(e, "ecc"); e->impl->set_replace_template(replace != 0); });
}
int gtx_ecc_exact_positions(gtx_ecc* e, int exact) {
  return guarded([&] { need(e, "ecc"); e->impl->set_exact_positions(exact != 0); });
}
int gtx_ecc_submit(gtx_ecc* e, const uint8_t* frame_bgr, int h, int w) {
  return guarded([&] { need(e, "ecc"); need(frame_bgr, "frame"); e->impl->submit_frame(frame_bgr, h, w); });
}
int gtx_ecc_submit_dev(gtx_ecc* e, gtx_ctx* producer, const void* frame_bgr_dptr, int h, int w) {
  return guarded([&] {
    need(e, "ecc"); need(frame_bgr_dptr, "frame");
    if (producer && producer->device != e->ctx->device) gtx::fail(-1, "gtx_ecc_submit_dev: the producer's context is on another device");
    e->impl->submit_frame_dev(frame_bgr_dptr, h, w, producer ? producer->stream : e->ctx->stream);
  });
}
int gtx_ecc_collect(gtx_ecc* e, double A[6], int info[2], double* rho) {
  return guarded([&] { need(e, "ecc"); need(A, "A"); e->impl->collect(A, info, rho); });
}
int gtx_ecc_image(gtx_ecc* e, int which, float* out) {
  return guarded([&] { need(e, "ecc"); need(out, "out"); e->impl->debug_image(which, out); });
}

struct gtx_sift {
  gtx_ctx* ctx;
  std::unique_ptr<gtx::Sift> impl;
};

int gtx_register_images(gtx_ctx* ctx, const gtx_reg_config* cfg, const uint8_t* src, int sh, int sw, const uint8_t* dst, int dh, int dw,
                        double H[9], int* valid, int stats[4], float timings_ms[4]) {
  return guarded([&] {
    need(ctx, "ctx"); need(cfg, "cfg"); need(src, "src"); need(dst, "dst"); need(H, "H"); need(valid, "valid"); need(stats, "stats");
    gtx::register_images(ctx, *cfg, src, sh, sw, dst, dh, dw, H, valid, stats, timings_ms);
  });
}
int gtx_sift_create(gtx_ctx* ctx, int max_h, int max_w, gtx_sift** out) {
  return guarded([&] {
    need(ctx, "ctx"); need(out, "out");
    std::unique_ptr<gtx_sift> s(new gtx_sift);
    s->ctx = ctx;
    s->impl.reset(new gtx::Sift(ctx->device, ctx->stream, max_h, max_w));
    *out = s.release();
  });
}
void gtx_sift_destroy(gtx_sift* s) { delete s; }
int gtx_sift_detect(gtx_sift* s, const uint8_t* image, int h, int w, int max_features, int root, float root_eps, int cap, int* n,
                    float* kp5, int* octave, float* desc) {
  return guarded([&] {
    need(s, "sift"); need(image, "image"); need(n, "n");
    s->impl->detect_and_compute(image, h, w, max_features, root != 0, root_eps);
    std::vector<gtx::SiftKeypoint> k;
    std::vector<float> d;
    s->impl->download(k, d);
    *n = (int)k.size();
    const int m = std::min<int>(cap, (int)k.size());
    for (int i = 0; i < m; ++i) {
      if (kp5) { kp5[5 * i] = k[i].x; kp5[5 * i + 1] = k[i].y; kp5[5 * i + 2] = k[i].size; kp5[5 * i + 3] = k[i].angle; kp5[5 * i + 4] = k[i].response; }
      if (octave) octave[i] = k[i].octave;
    }
    if (desc && m > 0) std::memcpy(desc, d.data(), (size_t)m * 128 * sizeof(float));
  });
}
int gtx_sift_stage_ms(gtx_sift* s, float out[4]) {
  return guarded([&] { need(s, "sift"); need(out, "out"); s->impl->stage_ms(out); });
}
int gtx_sift_pyramid(gtx_sift* s, int kind, int octave, int layer, int cap, float* out, int* h, int* w, int* n_octaves) {
  return guarded([&] {
    need(s, "sift"); need(h, "h"); need(w, "w");
    if (n_octaves) *n_octaves = s->impl->n_octaves();
    std::vector<float> img;
    s->impl->pyramid_image(kind, octave, layer, img, h, w);
    if (out) {
      if ((size_t)cap < img.size()) gtx::fail(GTX_ERR_INVALID, "pyramid image has %zu pixels, buffer holds %d", img.size(), cap);
      std::memcpy(out, img.data(), img.size() * sizeof(float));
    }
  });
}

int gtx_op_match_2nn(gtx_ctx* ctx, const float* query, int nq, const float* train, int nt, int* idx1, int* idx2, float* d1,
                     float* d2, int iters, float* ms_per_pass) {
  return guarded([&] {
    need(ctx, "ctx"); need(query, "query"); need(train, "train"); need(idx1, "idx1"); need(idx2, "idx2"); need(d1, "d1"); need(d2, "d2");
    if (nq < 0 || nt < 0) gtx::fail(GTX_ERR_INVALID, "negative descriptor count");
    GTX_HIP(hipSetDevice(ctx->device));
    if (nq == 0) return;
    hipStream_t s = ctx->stream;
    const size_t qn = (size_t)nq * 128, tn = (size_t)std::max(nt, 1) * 128;
    gtx::DevBuf qf(qn * 4), tf(tn * 4), qh(qn * 2), th(tn * 2), ws(gtx::match2nn_workspace_bytes(nq, nt));
    gtx::DevBuf i1(nq * 4), i2(nq * 4), e1(nq * 4), e2(nq * 4);
    GTX_HIP(hipMemcpy(qf.p, query, qn * 4, hipMemcpyHostToDevice));
    if (nt > 0) GTX_HIP(hipMemcpy(tf.p, train, (size_t)nt * 128 * 4, hipMemcpyHostToDevice));
    gtx::descriptors_to_half(qf.as<float>(), qh.p, qn, s);
    gtx::descriptors_to_half(tf.as<float>(), th.p, (size_t)nt * 128, s);
    gtx::match2nn(qh.p, qf.as<float>(), nq, th.p, tf.as<float>(), nt, ws.p, i1.as<int>(), i2.as<int>(), e1.as<float>(), e2.as<float>(), s);
    GTX_HIP(hipStreamSynchronize(s));
    if (iters > 0 && ms_per_pass) {
      hipEvent_t a, b;
      GTX_HIP(hipEventCreate(&a)); GTX_HIP(hipEventCreate(&b));
      GTX_HIP(hipEventRecord(a, s));
      for (int i = 0; i < iters; ++i)
        gtx::match2nn(qh.p, qf.as<float>(), nq, th.p, tf.as<float>(), nt, ws.p, i1.as<int>(), i2.as<int>(), e1.as<float>(), e2.as<float>(), s);
      GTX_HIP(hipEventRecord(b, s));
      GTX_HIP(hipEventSynchronize(b));
      float ms = 0.f;
      GTX_HIP(hipEventElapsedTime(&ms, a, b));
      *ms_per_pass = ms / iters;
      (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    GTX_HIP(hipMemcpy(idx1, i1.p, nq * 4, hipMemcpyDeviceToHost));
    GTX_HIP(hipMemcpy(idx2, i2.p, nq * 4, hipMemcpyDeviceToHost));
    GTX_HIP(hipMemcpy(d1, e1.p, nq * 4, hipMemcpyDeviceToHost));
    GTX_HIP(hipMemcpy(d2, e2.p, nq * 4, hipMemcpyDeviceToHost));
  });
}

int gtx_op_preprocess(gtx_ctx* ctx, int dtype, const uint8_t* frame, int h, int w, int net_h, int net_w,
                      void* out_img, uint8_t* out_gray, int gray_h, int gray_w) {
  return guarded([&] {
    need(ctx, "ctx"); need(frame, "frame"); need(out_img, "out_img");
    GTX_HIP(hipSetDevice(ctx->device));
    // Letterbox of the frame into exactly net_h x net_w (ultralytics geometry for that target).
    gtx::Letterbox lb{};
    lb.src_h = h; lb.src_w = w; lb.net_h = net_h; lb.net_w = net_w;
    const double r = std::min((double)net_h / h, (double)net_w / w);
    lb.new_w = (int)std::nearbyint(w * r);
    lb.new_h = (int)std::nearbyint(h * r);
    lb.top = (int)std::nearbyint((net_h - lb.new_h) / 2.0 - 0.1);
    lb.left = (int)std::nearbyint((net_w - lb.new_w) / 2.0 - 0.1);
    lb.gain = r;
    const size_t fb = (size_t)h * w * 3, npx = (size_t)net_h * net_w, ib = npx * 4;   // device image: RGB0 bytes
    gtx::DevBuf df(fb), di(ib), dg;
    if (out_gray) dg.alloc((size_t)gray_h * gray_w);
    GTX_HIP(hipMemcpy(df.p, frame, fb, hipMemcpyHostToDevice));
    gtx::launch_preprocess(dtype, df.as<uint8_t>(), 1, lb, di.p, out_gray ? dg.as<uint8_t>() : nullptr, gray_h, gray_w, ctx->stream);
    GTX_HIP(hipStreamSynchronize(ctx->stream));
    // The network's view of the image (what the stem kernels make of the bytes through their tables): byte / 255 in fp32,
    // rounded to fp16 for dtype f16.
    std::vector<uint8_t> raw(ib);
    GTX_HIP(hipMemcpy(raw.data(), di.p, ib, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < npx * 4; ++i) {
      const float f = (float)raw[i] / 255.f;
      if (dtype == gtx::DT_F16) static_cast<_Float16*>(out_img)[i] = (_Float16)f;
      else static_cast<float*>(out_img)[i] = f;
    }
    if (out_gray) GTX_HIP(hipMemcpy(out_gray, dg.p, (size_t)gray_h * gray_w, hipMemcpyDeviceToHost));
  });
}

/* ------------------------------------------------------------------ detector */

int gtx_detector_create(gtx_ctx* ctx, const gtx_det_config* cfg, gtx_detector** out) {
  return guarded([&] {
    need(ctx, "ctx"); need(cfg, "cfg"); need(out, "out");
    std::unique_ptr<gtx_detector> d(new gtx_detector);
    if (cfg->arch == 1) d->impl.reset(new gtx::RtDetr(ctx, *cfg));
    else if (cfg->arch == 0) d->impl.reset(new gtx::Detector(ctx, *cfg));
    else gtx::fail(-3, "gtx_det_config.arch %d: 0 (YOLOv8) or 1 (RT-DETR)", cfg->arch);
    *out = d.release();
  });
}
void gtx_detector_destroy(gtx_detector* det) { delete det; }

int gtx_detector_set_tensor(gtx_detector* det, const char* name, const float* data, int ndim, const int64_t* shape) {
  return guarded([&] {
    need(det, "det"); need(name, "name"); need(data, "data"); need(shape, "shape");
    det->impl->set_tensor(name, data, ndim, shape);
  });
}
int gtx_detector_finalize(gtx_detector* det) {
  return guarded([&] { need(det, "det"); det->impl->finalize(); });
}
int gtx_detector_input_size(gtx_detector* det, int* net_h, int* net_w) {
  return guarded([&] { need(det, "det"); need(net_h, "net_h"); need(net_w, "net_w"); det->impl->input_size(net_h, net_w); });
}
int gtx_detector_detect(gtx_detector* det, const uint8_t* frame_bgr, int h, int w, int* n_out, float* xyxy,
                        float* conf, int* cls, float speed_ms[3]) {
  return guarded([&] {
    need(det, "det"); need(frame_bgr, "frame"); need(n_out, "n_out"); need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls");
    det->impl->detect_host(frame_bgr, h, w, n_out, xyxy, conf, cls, speed_ms);
  });
}
int gtx_detector_detect_dev(gtx_detector* det, const void* frame_dptr, int h, int w, int* n_out, float* xyxy,
                            float* conf, int* cls, float speed_ms[3]) {
  return guarded([&] {
    need(det, "det"); need(frame_dptr, "frame"); need(n_out, "n_out"); need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls");
    det->impl->detect_dev(frame_dptr, 1, h, w, n_out, xyxy, conf, cls, speed_ms);
  });
}
int gtx_detector_detect_batch_dev(gtx_detector* det, const void* frames_dptr, int nb, int h, int w, int* n_out,
                                  float* xyxy, float* conf, int* cls, float speed_ms[3]) {
  return guarded([&] {
    need(det, "det"); need(frames_dptr, "frames"); need(n_out, "n_out"); need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls");
    det->impl->detect_dev(frames_dptr, nb, h, w, n_out, xyxy, conf, cls, speed_ms);
  });
}
int gtx_detector_submit_dev(gtx_detector* det, const void* frames_dptr, int nb, int h, int w) {
  return guarded([&] { need(det, "det"); need(frames_dptr, "frames"); det->impl->submit_dev(frames_dptr, nb, h, w); });
}
int gtx_detector_collect(gtx_detector* det, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) {
  return guarded([&] {
    need(det, "det"); need(n_out, "n_out"); need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls");
    det->impl->collect(n_out, xyxy, conf, cls, speed_ms);
  });
}
const void* gtx_detector_gray(gtx_detector* det, int b, int* gray_h, int* gray_w) {
  if (!det) return nullptr;
  return det->impl->gray(b, gray_h, gray_w);
}
int gtx_detector_raw_output(gtx_detector* det, int b, float* out, int* n_anchors) {
  return guarded([&] { need(det, "det"); need(out, "out"); det->impl->raw_output(b, out, n_anchors); });
}
int gtx_detector_raw_logits(gtx_detector* det, int b, float* out, int* n_anchors) {
  return guarded([&] { need(det, "det"); need(out, "out"); det->impl->raw_output(b, out, n_anchors, true); });
}
int gtx_detector_layer_output(gtx_detector* det, int b, const char* layer, float* out, int* h, int* w, int* c) {
  return guarded([&] { need(det, "det"); need(layer, "layer"); det->impl->layer_output(b, layer, out, h, w, c); });
}
int gtx_detector_saturated(gtx_detector* det, int clear, int* flag) {
  return guarded([&] { need(det, "det"); need(flag, "flag"); *flag = det->impl->saturated(clear != 0) ? 1 : 0; });
}
int gtx_detector_fell_back(gtx_detector* det, int* fell_back) {
  return guarded([&] { need(det, "det"); need(fell_back, "fell_back"); *fell_back = det->impl->fell_back() ? 1 : 0; });
}
int gtx_detector_pad_skip(gtx_detector* det, int* on, int* skipped, int* total) {
  return guarded([&] { need(det, "det"); det->impl->pad_skip(on, skipped, total); });
}
int gtx_detector_sparse_box(gtx_detector* det, int* on, int* overflows) {
  return guarded([&] { need(det, "det"); det->impl->sparse_box(on, overflows); });
}
int gtx_detector_features(gtx_detector* det, int b, float* out, int cap, int* n, int* dim) {
  return guarded([&] { need(det, "det"); det->impl->features(b, out, cap, n, dim); });
}
int gtx_detector_trace(gtx_detector* det, int every_n) {
  return guarded([&] { need(det, "det"); det->impl->set_trace(every_n); });
}
int gtx_detector_profile(gtx_detector* det, int nb, int iters, int cap, char* names, int* launches, float* total_ms,
                         double* flops, double* bytes, int* n_families) {
  return guarded([&] {
    need(det, "det"); need(names, "names"); need(launches, "launches"); need(total_ms, "total_ms");
    need(flops, "flops"); need(bytes, "bytes"); need(n_families, "n_families");
    std::vector<std::string> nm;
    std::vector<int> la;
    std::vector<float> ms;
    std::vector<double> fl, by;
    if (iters <= 0) det->impl->trace_report(nm, la, ms, fl, by);   // iters = 0: the live trace totals
    else det->impl->profile(nb, iters, nm, la, ms, fl, by);
    const int n = std::min<int>(cap, (int)nm.size());
    for (int i = 0; i < n; ++i) {
      std::strncpy(names + (size_t)i * 96, nm[i].c_str(), 95);
      names[(size_t)i * 96 + 95] = 0;
      launches[i] = la[i]; total_ms[i] = ms[i]; flops[i] = fl[i]; bytes[i] = by[i];
    }
    *n_families = n;
  });
}

/* ------------------------------------------------------------------ tracker */

int gtx_tracker_create(const gtx_tracker_config* cfg, gtx_tracker** out) {
  return guarded([&] {
    need(cfg, "cfg"); need(out, "out");
    std::unique_ptr<gtx_tracker> t(new gtx_tracker);
    if (cfg->type == 2 || cfg->type == 3) t->oc.reset(new gtx::OcSortTracker(*cfg));
    else if (cfg->type == 0 || cfg->type == 1 || cfg->type == 4) t->impl.reset(new gtx::ByteTracker(*cfg));
    else if (cfg->type == 5) t->tt.reset(new gtx::TrackTrackTracker(*cfg));
    else gtx::fail(GTX_ERR_INVALID, "tracker type %d (0 bytetrack, 1 botsort, 2 ocsort, 3 deepocsort, 4 fasttrack, 5 tracktrack)", cfg->type);
    *out = t.release();
  });
}
void gtx_tracker_destroy(gtx_tracker* trk) { delete trk; }
int gtx_tracker_reset(gtx_tracker* trk) {
  return guarded([&] { need(trk, "trk"); if (trk->oc) trk->oc->reset(); else if (trk->tt) trk->tt->reset(); else trk->impl->reset(); });
}
int gtx_tracker_update(gtx_tracker* trk, int n, const float* xyxy, const float* conf, const int* cls,
                       const double* gmc_affine, int cap, int* n_out, float* out_xyxy, int* out_id, float* out_score,
                       int* out_cls, int* out_det_idx) {
  return guarded([&] {
    need(trk, "trk"); need(n_out, "n_out");
    if (n > 0) { need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls"); }
    if (trk->oc) trk->oc->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx);
    else if (trk->tt) trk->tt->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx);
    else trk->impl->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx);
  });
}

int gtx_tracker_update_feats(gtx_tracker* trk, int n, const float* xyxy, const float* conf, const int* cls, const double* gmc_affine,
                             const float* feats, int feat_dim, int cap, int* n_out, float* out_xyxy, int* out_id, float* out_score,
                             int* out_cls, int* out_det_idx) {
  return guarded([&] {
    need(trk, "trk"); need(n_out, "n_out");
    if (n > 0) { need(xyxy, "xyxy"); need(conf, "conf"); need(cls, "cls"); }
    if (trk->oc) trk->oc->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx, feats, feat_dim);
    else if (trk->tt) trk->tt->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx, feats, feat_dim);
    else trk->impl->update(n, xyxy, conf, cls, gmc_affine, cap, n_out, out_xyxy, out_id, out_score, out_cls, out_det_idx, feats, feat_dim);
  });
}

int gtx_tracker_replay(gtx_tracker* trk, const double* recs, int n_recs, int stride, int max_det, int with_gmc, int row_cap,
                       int* rows_per_frame, float* row_xyxy, int* row_id, float* row_score, int* row_cls, int* row_det_idx) {
  return guarded([&] {
    need(trk, "trk"); need(rows_per_frame, "rows_per_frame");
    if (n_recs > 0) need(recs, "recs");
    if (row_cap > 0) {          // every row array is written unconditionally below
      need(row_xyxy, "row_xyxy"); need(row_id, "row_id"); need(row_score, "row_score"); need(row_cls, "row_cls"); need(row_det_idx, "row_det_idx");
    }
    const int tail = (with_gmc ? 7 : 0) + 10;
    if (max_det < 0 || stride != 1 + 6 * max_det + tail) gtx::fail(GTX_ERR_INVALID, "replay: stride %d does not match max_det %d", stride, max_det);
    std::vector<float> xyxy((size_t)4 * std::max(max_det, 1)), conf(std::max(max_det, 1));
    std::vector<int> cls(std::max(max_det, 1));
    int used = 0;
    for (int f = 0; f < n_recs; ++f) {
      const double* rec = recs + (size_t)f * stride;
      if (!std::isfinite(rec[0])) gtx::fail(GTX_ERR_INVALID, "replay: record %d has a non-finite detection count", f);
      const int n = (int)std::min(std::max(rec[0], 0.0), (double)max_det);
      for (int i = 0; i < n; ++i) {
        const double* d = rec + 1 + 6 * i;
        xyxy[4 * i] = (float)d[0]; xyxy[4 * i + 1] = (float)d[1]; xyxy[4 * i + 2] = (float)d[2]; xyxy[4 * i + 3] = (float)d[3];
        conf[i] = (float)d[4];
        cls[i] = (int)d[5];
      }
      const double* gmc = (with_gmc && rec[stride - 17] > 0) ? rec + stride - 16 : nullptr;
      int k = 0;
      const int room = row_cap - used;
      float* ox = row_xyxy + (size_t)4 * used;
      if (room <= 0 && n > 0) gtx::fail(GTX_ERR_INVALID, "replay: more than %d track rows", row_cap);
      if (trk->oc)
        trk->oc->update(n, xyxy.data(), conf.data(), cls.data(), gmc, room, &k, ox, row_id + used, row_score + used, row_cls + used, row_det_idx + used);
      else if (trk->tt)
        trk->tt->update(n, xyxy.data(), conf.data(), cls.data(), gmc, room, &k, ox, row_id + used, row_score + used, row_cls + used, row_det_idx + used);
      else
        trk->impl->update(n, xyxy.data(), conf.data(), cls.data(), gmc, room, &k, ox, row_id + used, row_score + used, row_cls + used, row_det_idx + used);
      rows_per_frame[f] = k;
      used += k;
    }
  });
}

int gtx_op_linear_assignment(const float* cost, int rows, int cols, double cost_limit, int* row_to_col, int* col_to_row) {
  return guarded([&] {
    if (rows < 0 || cols < 0) gtx::fail(GTX_ERR_INVALID, "linear assignment: negative size");
    if (rows > 0 && cols > 0) need(cost, "cost");
    if (rows > 0) need(row_to_col, "row_to_col");
    for (size_t i = 0; i < (size_t)rows * cols; ++i)
      if (!std::isfinite(cost[i])) gtx::fail(GTX_ERR_INVALID, "linear assignment: cost %zu is not finite", i);
    std::vector<int> x, y;
    if (cost_limit > 0 && std::isfinite(cost_limit)) {
      gtx::lap_limited(cost, rows, cols, cost_limit, x, y);
    } else {
      std::vector<double> c((size_t)rows * cols);
      for (size_t i = 0; i < c.size(); ++i) c[i] = cost[i];
      gtx::lap_full(c, rows, cols, x);
      y.assign(cols, -1);
      for (int r = 0; r < rows; ++r)
        if (x[r] >= 0) y[x[r]] = r;
    }
    for (int r = 0; r < rows; ++r) row_to_col[r] = x[r];
    if (col_to_row)
      for (int c = 0; c < cols; ++c) col_to_row[c] = y[c];
  });
}

/* ------------------------------------------------------------------ stabilizer */

int gtx_stabilizer_create(gtx_ctx* ctx, const gtx_stab_config* cfg, gtx_stabilizer** out) {
  return guarded([&] {
    need(ctx, "ctx"); need(cfg, "cfg"); need(out, "out");
    std::unique_ptr<gtx_stabilizer> s(new gtx_stabilizer);
    s->impl.reset(new gtx::Stabilizer(ctx, *cfg));
    *out = s.release();
  });
}
void gtx_stabilizer_destroy(gtx_stabilizer* st) { delete st; }
int gtx_stabilizer_set_ref_frame(gtx_stabilizer* st, const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n) {
  return guarded([&] { need(st, "st"); need(frame_bgr, "frame"); st->impl->set_ref_frame(frame_bgr, h, w, boxes_xywh, n); });
}
int gtx_stabilizer_set_ref_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw, const float* boxes_xywh, int n) {
  return guarded([&] { need(st, "st"); need(gray_dptr, "gray"); st->impl->set_ref_gray_dev(gray_dptr, gh, gw, boxes_xywh, n); });
}
int gtx_stabilizer_stabilize(gtx_stabilizer* st, const uint8_t* frame_bgr, int h, int w, const float* boxes_xywh, int n,
                             double H[9], int* valid, int stats[4]) {
  return guarded([&] {
    need(st, "st"); need(frame_bgr, "frame"); need(H, "H"); need(valid, "valid");
    st->impl->stabilize(frame_bgr, h, w, boxes_xywh, n, H, valid, stats);
  });
}
int gtx_stabilizer_stabilize_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw, const float* boxes_xywh,
                                      int n, double H[9], int* valid, int stats[4]) {
  return guarded([&] {
    need(st, "st"); need(gray_dptr, "gray"); need(H, "H"); need(valid, "valid");
    st->impl->stabilize_gray_dev(gray_dptr, gh, gw, boxes_xywh, n, H, valid, stats);
  });
}
int gtx_stabilizer_promote_cur(gtx_stabilizer* st) {
  return guarded([&] { need(st, "st"); st->impl->promote_cur_to_ref(); });
}
int gtx_stabilizer_submit_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw, const float* boxes_xywh, int n) {
  return guarded([&] { need(st, "st"); need(gray_dptr, "gray"); st->impl->submit_gray_dev(gray_dptr, gh, gw, boxes_xywh, n); });
}
int gtx_stabilizer_collect(gtx_stabilizer* st, double H[9], int* valid, int stats[4]) {
  return guarded([&] { need(st, "st"); need(H, "H"); need(valid, "valid"); st->impl->collect(H, valid, stats); });
}
int gtx_stabilizer_keypoints(gtx_stabilizer* st, int which, int cap, int* n, float* xy, int* level, int* angle_bin, uint8_t* desc) {
  return guarded([&] { need(st, "st"); need(n, "n"); st->impl->keypoints(which, cap, n, xy, level, angle_bin, desc); });
}
int gtx_stabilizer_matches(gtx_stabilizer* st, int cap, int* n, int* cur_idx, int* ref_idx, int* dist) {
  return guarded([&] { need(st, "st"); need(n, "n"); st->impl->matches(cap, n, cur_idx, ref_idx, dist); });
}

int gtx_stabilizer_last_ms(gtx_stabilizer* st, float* ms) {
  return guarded([&] { need(st, "st"); need(ms, "ms"); *ms = st->impl->last_ms(); });
}

int gtx_stabilizer_pattern(gtx_stabilizer* st, int8_t* out) {
  return guarded([&] {
    need(out, "out");
    if (st) { st->impl->pattern(out); return; }
    std::vector<int8_t> t;                        // no object: the built-in table (host only, needs no device)
    gtx::stabilizer_pattern_table(t);
    std::memcpy(out, t.data(), t.size());
  });
}

/* ------------------------------------------------------------------ geometry */

int gtx_warp_boxes(const double H[9], const float* xywh_in, int n, float* xywh_out) {
  return guarded([&] {
    need(H, "H");
    if (n > 0) { need(xywh_in, "xywh_in"); need(xywh_out, "xywh_out"); }
    gtx::warp_boxes(H, xywh_in, n, xywh_out);
  });
}
int gtx_perspective_points(const double H[9], const double* x, const double* y, int n, double* ox, double* oy) {
  return guarded([&] {
    need(H, "H");
    if (n > 0) { need(x, "x"); need(y, "y"); need(ox, "ox"); need(oy, "oy"); }
    gtx::perspective_points(H, x, y, n, ox, oy);
  });
}

int gtx_op_estimate_affine_partial(const float* p_xy, const float* q_xy, int n, unsigned seed, double A[6], int* valid, int* n_inliers) {
  return guarded([&] {
    need(A, "A"); need(valid, "valid");
    if (n > 0) { need(p_xy, "p_xy"); need(q_xy, "q_xy"); }
    if (n < 0) gtx::fail(GTX_ERR_INVALID, "estimate_affine_partial: n = %d", n);
    *valid = gtx::estimate_affine_partial(p_xy, q_xy, n, seed, A, n_inliers) ? 1 : 0;
  });
}
int gtx_op_georef_points(gtx_ctx* ctx, const gtx_georef_chain* chain, const double* x, const double* y, int n,
                         double* ortho_x, double* ortho_y, double* lat, double* lon, double* east, double* north) {
  return guarded([&] {
    need(ctx, "ctx"); need(chain, "chain");
    if (n > 0) { need(x, "x"); need(y, "y"); }
    gtx::georef_points(ctx, *chain, x, y, n, ortho_x, ortho_y, lat, lon, east, north);
  });
}
int gtx_yuv420_to_bgr_dev(gtx_ctx* ctx, const void* yuv_dptr, int h, int w, void* bgr_dptr) {
  return guarded([&] {
    need(ctx, "ctx"); need(yuv_dptr, "yuv"); need(bgr_dptr, "bgr");
    gtx::yuv420_to_bgr_dev(ctx, yuv_dptr, h, w, bgr_dptr);
  });
}

int gtx_warp_frame_dev(gtx_ctx* ctx, const void* src_dptr, int h, int w, const double H[9], void* dst_dptr) {
  return guarded([&] {
    need(ctx, "ctx"); need(src_dptr, "src"); need(H, "H"); need(dst_dptr, "dst");
    if (src_dptr == dst_dptr) gtx::fail(GTX_ERR_INVALID, "warp_frame: source and destination must be distinct buffers");
    gtx::warp_frame_dev(ctx, src_dptr, h, w, H, dst_dptr);
  });
}

int gtx_op_clahe(gtx_ctx* ctx, const uint8_t* gray, int h, int w, uint8_t* out) {
  return guarded([&] {
    need(ctx, "ctx"); need(gray, "gray"); need(out, "out");
    gtx::clahe_image(ctx, gray, h, w, out);
  });
}

int gtx_warp_frame(gtx_ctx* ctx, const uint8_t* src_bgr, int h, int w, const double H[9], uint8_t* dst_bgr) {
  return guarded([&] {
    need(ctx, "ctx"); need(src_bgr, "src"); need(H, "H"); need(dst_bgr, "dst");
    gtx::warp_frame(ctx, src_bgr, h, w, H, dst_bgr);
  });
}

}  // extern "C"
