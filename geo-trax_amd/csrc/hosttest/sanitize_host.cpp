// Host-only driver for the sanitizer builds (make -C geo-trax_amd asan | tsan): the tracker and the projective-geometry
// helpers are the parts of libgtx that run on the host inside the hot loop (the reference has no
// sanitizer story; a multi-threaded, multi-stream build should own one). GPU AddressSanitizer is not available on this
// pool, so the HIP kernels are out of reach of this target; the C++ that manages tracks, lists and the LAP is not.
//   asan: one long seeded detection stream through ByteTrack and BoT-SORT (births, deaths, empty frames, dense overlaps,
//         >1000 removed tracks so that the ring of removed ids wraps), box warps and point transforms on edge inputs.
//   tsan: the same streams on several tracker objects from several threads at once (objects are independent; the library
//         promises nothing shared between them).
//   both: the result-file writers, whose worker threads format chunks and pass the file to each other in row order.
#include <cmath>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

#include <cstdint>
#include <string>

#include "../../../include/gtx.h"
#include "../geometry.hpp"
#include "../tracker.hpp"
#include "../vec_acos.hpp"

namespace {
struct Frame { std::vector<float> xyxy, conf; std::vector<int> cls; };

std::vector<Frame> stream(unsigned seed, int n_obj, int n_frames, float w, float h) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::vector<float> px(n_obj), py(n_obj), vx(n_obj), vy(n_obj), sw(n_obj), sh(n_obj), cf(n_obj);
  for (int k = 0; k < n_obj; ++k) {
    px[k] = 50 + U(rng) * (w - 100); py[k] = 50 + U(rng) * (h - 100); vx[k] = (U(rng) - 0.5f) * 12; vy[k] = (U(rng) - 0.5f) * 12;
    sw[k] = 40 + U(rng) * 120; sh[k] = 25 + U(rng) * 45; cf[k] = 0.3f + U(rng) * 0.65f;
  }
  std::vector<Frame> out(n_frames);
  for (int t = 0; t < n_frames; ++t) {
    if (t % 37 == 17) continue;                       // an empty frame now and then
    for (int k = 0; k < n_obj; ++k) {
      if (U(rng) < 0.1f) continue;
      if ((t + k) % 61 == 0) { px[k] = 50 + U(rng) * (w - 100); py[k] = 50 + U(rng) * (h - 100); }   // teleport: old track dies, new one is born
      const float cx = px[k] + vx[k] * (t % 40), cy = py[k] + vy[k] * (t % 40);
      out[t].xyxy.insert(out[t].xyxy.end(), {cx - sw[k] / 2, cy - sh[k] / 2, cx + sw[k] / 2, cy + sh[k] / 2});
      out[t].conf.push_back(U(rng) < 0.1f ? 0.12f + 0.1f * U(rng) : cf[k]);
      out[t].cls.push_back(k % 4);
    }
  }
  return out;
}

long run_tracker(int type, unsigned seed) {
  gtx_tracker_config cfg{};
  cfg.type = type; cfg.track_high_thresh = 0.25f; cfg.track_low_thresh = 0.1f; cfg.new_track_thresh = 0.25f; cfg.track_buffer = 30;
  cfg.match_thresh = 0.8f; cfg.fuse_score = 1; cfg.frame_rate = 30;
  gtx::ByteTracker trk(cfg);
  const int cap = 512;
  std::vector<float> ob(cap * 4), os(cap);
  std::vector<int> oi(cap), oc(cap), od(cap);
  long rows = 0;
  for (int rep = 0; rep < 2; ++rep) {
    for (const Frame& f : stream(seed, rep ? 150 : 60, 400, rep ? 900.f : 3840.f, rep ? 600.f : 2160.f)) {   // second pass: dense overlaps
      const double warp[6] = {1.0, 1e-4, 0.3, -1e-4, 1.0, -0.2};
      int n = 0;
      trk.update((int)f.conf.size(), f.xyxy.data(), f.conf.data(), f.cls.data(), type == 1 ? warp : nullptr, cap, &n, ob.data(), oi.data(),
                 os.data(), oc.data(), od.data());
      rows += n;
    }
    if (rep == 0) trk.reset();
  }
  return rows;
}

// Round 6: the trackers with an appearance branch (Deep OC-SORT, TrackTrack; BoT-SORT's is in run_tracker's type 1 when feats are
// handed in) on the same streams with one seeded vector per detection, a camera warp, and the partial-affine fit of the feature GMC.
template <class Tracker>
long run_reid_tracker(int type, unsigned seed) {
  gtx_tracker_config cfg{};
  cfg.type = type; cfg.track_high_thresh = 0.3f; cfg.track_low_thresh = 0.1f; cfg.new_track_thresh = 0.3f; cfg.track_buffer = 30;
  cfg.match_thresh = 0.8f; cfg.fuse_score = 1; cfg.frame_rate = 30; cfg.delta_t = 3; cfg.inertia = 0.2f; cfg.min_hits = 3;
  cfg.with_reid = 1; cfg.proximity_thresh = 0.3f; cfg.appearance_thresh = 0.5f; cfg.alpha_fixed_emb = 0.9f;
  cfg.iou_weight = 0.5f; cfg.reid_weight = 0.5f; cfg.conf_weight = 0.1f; cfg.angle_weight = 0.05f; cfg.penalty_p = 0.2f; cfg.penalty_q = 0.4f;
  cfg.reduce_step = 0.05f; cfg.tai_thr = 0.55f; cfg.min_track_len = 3; cfg.lost_match_thr = 0.9f;
  Tracker trk(cfg);
  const int cap = 512, dim = 24;
  std::vector<float> ob(cap * 4), os(cap);
  std::vector<int> oi(cap), oc(cap), od(cap);
  std::mt19937 rng(seed * 7 + 1);
  std::normal_distribution<float> N(0.f, 1.f);
  long rows = 0;
  for (const Frame& f : stream(seed, 80, 300, 1920.f, 1080.f)) {
    const int n_det = (int)f.conf.size();
    std::vector<float> feats((size_t)n_det * dim);
    for (int i = 0; i < n_det; ++i)
      for (int k = 0; k < dim; ++k) feats[(size_t)i * dim + k] = (float)((f.cls[i] + 1) * ((k % 5) - 2)) + 0.4f * N(rng);
    const double warp[6] = {1.0, 2e-4, 0.5, -2e-4, 1.0, -0.3};
    int n = 0;
    trk.update(n_det, f.xyxy.data(), f.conf.data(), f.cls.data(), warp, cap, &n, ob.data(), oi.data(), os.data(), oc.data(), od.data(),
               n_det ? feats.data() : nullptr, dim);
    rows += n;
  }
  return rows;
}

// vec_acos.hpp against libm over the whole argument range, the range borders and the neighbourhoods of +-1 (where acos is
// steepest): at most 2 ulp of pi apart, 0 and pi exactly at the ends.
bool run_acos_check(unsigned seed) {
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const int n = 1 << 18;
  std::vector<double> x(n), y(n);
  for (int i = 0; i < n; ++i) x[i] = U(g);
  const double edge[] = {1.0, -1.0, 0.0, -0.0, 0.5, -0.5, 0.49999999999999994, -0.49999999999999994, 1.0 - 1e-16, -1.0 + 1e-16, 1e-300, -1e-300};
  int k = 0;
  for (double e : edge) x[k++] = e;
  for (int i = 0; i < 4000; ++i) { x[k++] = std::cos(i * 1e-9); x[k++] = -std::cos(i * 1e-9); }
  gtx::acos_block(x.data(), y.data(), n);
  if (y[0] != 0.0 || y[1] != M_PI) return false;
  for (int i = 0; i < n; ++i)
    if (!(std::fabs(y[i] - std::acos(x[i])) <= 9e-16)) { std::fprintf(stderr, "acos_block(%.17g) = %.17g, libm %.17g\n", x[i], y[i], std::acos(x[i])); return false; }
  return true;
}

bool run_affine_fit(unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(0.f, 1000.f);
  for (int n : {0, 1, 2, 5, 400}) {
    std::vector<float> p((size_t)2 * n), q((size_t)2 * n);
    for (int i = 0; i < n; ++i) {
      p[2 * i] = U(rng); p[2 * i + 1] = U(rng);
      q[2 * i] = 1.001f * p[2 * i] - 0.002f * p[2 * i + 1] + 3.f + (i % 4 == 0 ? U(rng) * 0.3f : 0.f);     // a quarter of the pairs are outliers
      q[2 * i + 1] = 0.002f * p[2 * i] + 1.001f * p[2 * i + 1] - 2.f;
    }
    double A[6];
    int inl = 0;
    const bool ok = gtx::estimate_affine_partial(p.data(), q.data(), n, seed, A, &inl);
    if (n >= 5 && (!ok || std::fabs(A[0] - 1.001) > 0.01 || inl < n / 2)) return false;
    if (n < 2 && ok) return false;
  }
  return true;
}
}  // namespace

// The result-file writers (table_writer.cpp): several chunks per call so that their own threads hand the file to each other in
// order, awkward values, a categorical column with missing cells; the anchor walk over ragged tracks incl. an empty one.
bool run_writers(unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> U(-4000.f, 4000.f);
  const int64_t rows = 40000;
  std::vector<float> t((size_t)rows * 12);
  for (float& v : t) v = U(rng);
  t[5] = NAN; t[17] = INFINITY; t[29] = -0.0f; t[41] = 1e-38f;
  const std::string base = "/tmp/gtx_sanitize_" + std::to_string(seed);
  if (gtx_write_table_f32((base + ".txt").c_str(), t.data(), rows, 12, 6, 4) != GTX_OK) return false;
  std::vector<double> d(t.begin(), t.begin() + 9 * 1000);
  if (gtx_write_table_f64((base + "_h.txt").c_str(), d.data(), 1000, 9, 20, 3) != GTX_OK) return false;
  std::vector<int64_t> ids((size_t)rows);
  std::vector<double> val((size_t)rows);
  std::vector<int32_t> code((size_t)rows);
  for (int64_t i = 0; i < rows; ++i) { ids[i] = i / 7 - 3; val[i] = i % 11 ? (double)t[i] * 1e-3 : (double)NAN; code[i] = (int32_t)(i % 4) - 1; }
  const char* labels[3] = {"A", "\"B,1\"", ""};
  const int kinds[3] = {0, 1, 2};
  const void* cols[3] = {ids.data(), val.data(), code.data()};
  const char* const* cats[3] = {nullptr, nullptr, labels};
  const int ncat[3] = {0, 0, 3};
  if (gtx_write_csv((base + ".csv").c_str(), "Vehicle_ID,Value,Section", 3, kinds, cols, cats, ncat, rows, 4) != GTX_OK) return false;
  if (gtx_write_csv("/no/such/dir/x.csv", "a,b,c", 3, kinds, cols, cats, ncat, rows, 2) == GTX_OK) return false;   // the error path closes nothing twice
  std::vector<int64_t> start = {0, 0, 1, 300, 2000, 2001, (int64_t)rows};
  std::vector<uint8_t> anchor((size_t)rows, 7);
  std::vector<float> dx((size_t)rows), dy((size_t)rows);
  if (gtx_track_anchor_walk(t.data(), t.data() + rows, start.data(), (int)start.size() - 1, 45.87156f, anchor.data(), dx.data(), dy.data()) != GTX_OK) return false;
  std::remove((base + ".txt").c_str()); std::remove((base + "_h.txt").c_str()); std::remove((base + ".csv").c_str());
  return anchor[0] == 0 && anchor[1] <= 1;
}

int main(int argc, char** argv) {
  const bool threads = argc > 1 && argv[1][0] == 't';
  long rows = 0;
  if (!run_writers(threads ? 2 : 1)) { std::fprintf(stderr, "result-file writers failed\n"); return 3; }
  if (threads) {
    std::vector<long> r(6);
    std::vector<std::thread> th;
    for (int i = 0; i < 6; ++i)
      th.emplace_back([&r, i] {
        r[i] = i < 4 ? run_tracker(i & 1, 100 + i) : (i == 4 ? run_reid_tracker<gtx::OcSortTracker>(3, 104) : run_reid_tracker<gtx::TrackTrackTracker>(5, 105));
      });
    for (auto& t : th) t.join();
    for (long v : r) rows += v;
  } else {
    rows = run_tracker(0, 1) + run_tracker(1, 2);
    rows += run_reid_tracker<gtx::OcSortTracker>(3, 3) + run_reid_tracker<gtx::TrackTrackTracker>(5, 4);
    if (!run_affine_fit(5)) { std::fprintf(stderr, "partial-affine fit wrong\n"); return 4; }
    if (!run_acos_check(6)) { std::fprintf(stderr, "vector acos differs from libm\n"); return 5; }
    // geometry helpers on edge inputs: zero boxes, a point on the line at infinity, a degenerate matrix
    const double H[9] = {1.01, 0.002, 3.0, -0.001, 0.99, -6.0, 1e-7, -1e-7, 1.0};
    std::vector<float> in = {100, 200, 50, 20, 3800, 2100, 90, 40}, outb(8);
    gtx::warp_boxes(H, in.data(), 2, outb.data());
    gtx::warp_boxes(H, in.data(), 0, outb.data());
    const double Hinf[9] = {1, 0, 0, 0, 1, 0, 1, 0, -5};
    double x[2] = {5.0, 1.0}, y[2] = {7.0, 2.0}, ox[2], oy[2];
    gtx::perspective_points(Hinf, x, y, 2, ox, oy);
    if (!(ox[0] == 0.0 && oy[0] == 0.0) || !std::isfinite(outb[0])) { std::fprintf(stderr, "geometry edge cases wrong\n"); return 2; }
  }
  std::printf("sanitize_host ok: %ld track rows\n", rows);
  return rows > 10000 ? 0 : 1;
}
