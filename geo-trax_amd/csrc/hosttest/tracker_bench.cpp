// Host-only timing of the trackers' update() on a seeded stream of 132 moving boxes per 3840x2160 frame (the golden clip's
// box count): `make -C geo-trax_amd trackerbench`. What the serial tracker stage of the extract loop -- and rank 0's replay
// of a frame-sharded run -- costs per frame, without ctypes and without a GPU. DESIGN.md section 6 quotes these numbers.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../tracker.hpp"

namespace {
gtx_tracker_config config_of(int type) {
  gtx_tracker_config c;
  std::memset(&c, 0, sizeof c);
  c.type = type;
  c.track_high_thresh = 0.25f; c.track_low_thresh = 0.1f; c.new_track_thresh = 0.25f; c.track_buffer = 30; c.match_thresh = 0.8f;
  c.fuse_score = 1; c.frame_rate = 30; c.delta_t = 3; c.inertia = 0.2f; c.use_byte = 0; c.min_hits = 3;
  c.reset_velocity_offset_occ = 5; c.reset_pos_offset_occ = 3; c.enlarge_bbox_occ = 1.1f; c.dampen_motion_occ = 0.5f;
  c.active_occ_to_lost_thresh = 10; c.occ_cover_thresh = 0.7f; c.occ_reappear_window = 40; c.init_iou_suppress = 0.7f;
  c.proximity_thresh = 0.5f; c.appearance_thresh = 0.8f; c.iou_weight = 0.5f; c.reid_weight = 0.5f; c.conf_weight = 0.1f;
  c.angle_weight = 0.05f; c.penalty_p = 0.2f; c.penalty_q = 0.4f; c.reduce_step = 0.05f; c.tai_thr = 0.55f; c.min_track_len = 3;
  c.alpha_fixed_emb = 0.95f;
  return c;
}

template <class T>
double run(int type, int frames, int n, int* tracks) {
  T t(config_of(type));
  std::mt19937 g(0);
  std::uniform_real_distribution<float> U(0, 1);
  std::normal_distribution<float> N(0, 1);
  std::vector<float> cx(n), cy(n), w(n), h(n), vx(n), vy(n), b(n * 4), conf(n), ox(4096 * 4), os(4096);
  std::vector<int> cls(n, 0), oid(4096), oc(4096), oi(4096);
  for (int i = 0; i < n; ++i) { cx[i] = 100 + 3600 * U(g); cy[i] = 100 + 1900 * U(g); w[i] = 30 + 50 * U(g); h[i] = 30 + 50 * U(g); vx[i] = 2 * N(g); vy[i] = 2 * N(g); }
  double tot = 0;
  for (int f = 0; f < frames; ++f) {
    for (int i = 0; i < n; ++i) {
      const float x = cx[i] + vx[i] * (f % 300) + 0.5f * N(g), y = cy[i] + vy[i] * (f % 300) + 0.5f * N(g);
      b[i * 4] = x - w[i] / 2; b[i * 4 + 1] = y - h[i] / 2; b[i * 4 + 2] = x + w[i] / 2; b[i * 4 + 3] = y + h[i] / 2;
      conf[i] = 0.3f + 0.6f * U(g);
    }
    const auto t0 = std::chrono::steady_clock::now();
    t.update(n, b.data(), conf.data(), cls.data(), nullptr, 4096, tracks, ox.data(), oid.data(), os.data(), oc.data(), oi.data());
    tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return 1e6 * tot / frames;
}
}  // namespace

int main(int argc, char** argv) {
  const int frames = argc > 1 ? std::atoi(argv[1]) : 2000, n = argc > 2 ? std::atoi(argv[2]) : 132;
  const char* names[6] = {"bytetrack", "botsort", "ocsort", "deepocsort", "fasttrack", "tracktrack"};
  for (int type = 0; type < 6; ++type) {
    int tracks = 0;
    const double us = (type == 2 || type == 3) ? run<gtx::OcSortTracker>(type, frames, n, &tracks)
                      : type == 5             ? run<gtx::TrackTrackTracker>(type, frames, n, &tracks)
                                              : run<gtx::ByteTracker>(type, frames, n, &tracks);
    std::printf("%-11s %8.1f us per frame (%d boxes per frame, %d tracks at the end)\n", names[type], us, n, tracks);
  }
  return 0;
}
