#include "register.hpp"

#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "detector.hpp"
#include "match_l2.hpp"
#include "sift.hpp"
#include "stabilizer.hpp"

namespace gtx {

namespace {
// The pyramids of a 4K image are ~2 GB: allocating and freeing them per call costs far more than the
// registration itself. Keep the last few Sift objects (keyed by context and image size) alive.
struct SiftCache {
  struct Entry { gtx_ctx* ctx; int h, w; std::unique_ptr<Sift> sift; unsigned long stamp; };
  std::vector<Entry> entries;
  unsigned long clock = 0;
  std::mutex mu;
  Sift* get(gtx_ctx* ctx, int h, int w, const Sift* other) {
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : entries)
      if (e.ctx == ctx && e.h == h && e.w == w && e.sift.get() != other) { e.stamp = ++clock; return e.sift.get(); }
    if (entries.size() >= 4) {
      size_t victim = entries.size();
      for (size_t i = 0; i < entries.size(); ++i)
        if (entries[i].sift.get() != other && (victim == entries.size() || entries[i].stamp < entries[victim].stamp)) victim = i;
      entries.erase(entries.begin() + (long)victim);
    }
    entries.push_back(Entry{ctx, h, w, std::unique_ptr<Sift>(new Sift(ctx->device, ctx->stream, h, w)), ++clock});
    return entries.back().sift.get();
  }
};
SiftCache& sift_cache() {
  static SiftCache c;
  return c;
}
}  // namespace

void register_images(gtx_ctx* ctx, const gtx_reg_config& cfg, const uint8_t* src, int sh, int sw, const uint8_t* dst, int dh, int dw,
                     double H[9], int* valid, int stats[4], float timings_ms[4]) {
  GTX_CHECK(cfg.max_features >= 4, "registration: max_features=%d", cfg.max_features);
  GTX_CHECK(cfg.filter_ratio > 0.f && cfg.filter_ratio <= 1.f, "registration: filter_ratio=%g outside (0,1]", cfg.filter_ratio);
  GTX_CHECK(cfg.ransac_threshold > 0.f, "registration: ransac threshold must be positive");
  GTX_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t0) { return std::chrono::duration<float, std::milli>(clk::now() - t0).count(); };
  *valid = 0;
  std::memset(stats, 0, sizeof(int) * 4);
  float tm[4] = {0, 0, 0, 0};

  auto t0 = clk::now();
  Sift& sift_dst = *sift_cache().get(ctx, dh, dw, nullptr);
  Sift& sift_src = *sift_cache().get(ctx, sh, sw, &sift_dst);
  const bool root = cfg.rsift_eps >= 0.f;        // rsift_eps < 0: plain SIFT descriptors (stabilo `detector_name: sift`, default.yaml:109)
  sift_dst.detect_and_compute(dst, dh, dw, cfg.max_features, root, cfg.rsift_eps);     // reference = destination
  sift_src.detect_and_compute(src, sh, sw, cfg.max_features, root, cfg.rsift_eps);     // current = source (query)
  GTX_HIP(hipStreamSynchronize(s));
  tm[0] = ms_since(t0);
  const int nq = sift_src.count(), nt = sift_dst.count();
  stats[0] = nq; stats[1] = nt;
  if (nq >= 1 && nt >= 2) {
    t0 = clk::now();
    DevBuf qh((size_t)nq * 256), th((size_t)nt * 256), ws(match2nn_workspace_bytes(nq, nt));
    DevBuf i1(nq * 4), i2(nq * 4), d1(nq * 4), d2(nq * 4);
    descriptors_to_half(sift_src.descriptors_dev(), qh.p, (size_t)nq * 128, s);
    descriptors_to_half(sift_dst.descriptors_dev(), th.p, (size_t)nt * 128, s);
    match2nn(qh.p, sift_src.descriptors_dev(), nq, th.p, sift_dst.descriptors_dev(), nt, ws.p, i1.as<int>(), i2.as<int>(), d1.as<float>(),
             d2.as<float>(), s);
    std::vector<int> h1(nq), h2(nq);
    std::vector<float> e1(nq), e2(nq);
    GTX_HIP(hipMemcpyAsync(h1.data(), i1.p, nq * 4, hipMemcpyDeviceToHost, s));
    GTX_HIP(hipMemcpyAsync(h2.data(), i2.p, nq * 4, hipMemcpyDeviceToHost, s));
    GTX_HIP(hipMemcpyAsync(e1.data(), d1.p, nq * 4, hipMemcpyDeviceToHost, s));
    GTX_HIP(hipMemcpyAsync(e2.data(), d2.p, nq * 4, hipMemcpyDeviceToHost, s));
    GTX_HIP(hipStreamSynchronize(s));
    tm[1] = ms_since(t0);
    // Lowe ratio in query order (stabilo: m.distance < ratio * n.distance)
    t0 = clk::now();
    const std::vector<SiftKeypoint>&kq = sift_src.keypoints_host(), &kt = sift_dst.keypoints_host();   // positions only: the descriptors stay in HBM
    std::vector<float4> pts;
    for (int i = 0; i < nq; ++i)
      if (h1[i] >= 0 && h2[i] >= 0 && e1[i] < cfg.filter_ratio * e2[i]) pts.push_back(make_float4(kq[i].x, kq[i].y, kt[h1[i]].x, kt[h1[i]].y));
    stats[2] = (int)pts.size();
    tm[2] = ms_since(t0);
    if (pts.size() >= 4) {
      t0 = clk::now();
      DevBuf dp(sizeof(float4) * pts.size());
      GTX_HIP(hipMemcpyAsync(dp.p, pts.data(), sizeof(float4) * pts.size(), hipMemcpyHostToDevice, s));
      const int n_hyp = std::max(256, std::min(cfg.ransac_max_iter, 16384));
      int n_inl = 0;
      if (ransac_homography(ctx->device, s, dp.as<float4>(), (int)pts.size(), (unsigned)cfg.seed, n_hyp, dw, dh, cfg.ransac_threshold, H, &n_inl)) {
        *valid = 1;
        stats[3] = n_inl;
      }
      tm[3] = ms_since(t0);
    }
  }
  if (timings_ms) std::memcpy(timings_ms, tm, sizeof tm);
}

}  // namespace gtx
