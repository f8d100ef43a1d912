// YOLOv8 (detect) graph builder + executor. Layer topology follows ultralytics'
// cfg/models/v8/yolov8.yaml (backbone 0-9, head 10-22); channel widths and bottleneck counts
// are read off the tensor shapes, so every v8 scale (n/s/m/l/x) loads unchanged.
#include "detector.hpp"
#include "split_format.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <deque>
#include <functional>
#include <mutex>

namespace gtx {

Detector::Detector(gtx_ctx* ctx, const gtx_det_config& cfg) : ctx_(ctx), cfg_(cfg) {
  GTX_CHECK(cfg.imgsz > 0 && cfg.imgsz % 32 == 0, "imgsz must be a positive multiple of 32 (got %d)", cfg.imgsz);
  GTX_CHECK(cfg.max_det > 0 && cfg.nc > 0 && cfg.nc <= 128, "max_det must be positive and nc in [1, 128] (got %d, %d)", cfg.max_det, cfg.nc);
  GTX_CHECK(cfg.frame_h > 0 && cfg.frame_w > 0, "frame size must be given");
  if (cfg_.max_batch < 1) cfg_.max_batch = 1;
  dtype_ = cfg.half ? DT_F16 : DT_F32;
  conv_dtype_ = (!cfg.half && cfg.fp32_split) ? DT_F32S : dtype_;
  es_ = dtype_size(dtype_);
  lb_ = letterbox_geometry(cfg.frame_h, cfg.frame_w, cfg.imgsz, cfg.rect != 0, 32);
  GTX_CHECK(lb_.net_h % 32 == 0 && lb_.net_w % 32 == 0, "network input %dx%d is not stride aligned", lb_.net_h, lb_.net_w);
  GTX_HIP(hipSetDevice(ctx->device));
  for (auto& e : ev_) GTX_HIP(hipEventCreateWithFlags(&e, wait_event_flags(true)));
  for (auto& e : ev_up_) GTX_HIP(hipEventCreate(&e));
}

Detector::~Detector() {
  if (h_out_n_) (void)hipHostFree(h_out_n_);
  if (h_sat_) (void)hipHostFree(h_sat_);
  if (h_out_rows_) (void)hipHostFree(h_out_rows_);
  if (h_feats_) (void)hipHostFree(h_feats_);
  if (h_count_) (void)hipHostFree(h_count_);
  for (auto& e : ev_)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_up_)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : trace_ev_)
    if (e) (void)hipEventDestroy(e);
  if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
}

void Detector::set_tensor(const std::string& name, const float* data, int ndim, const int64_t* shape) {
  GTX_CHECK(!finalized_, "set_tensor after finalize");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.data.assign(data, data + n);
  tensors_[name] = std::move(t);
}

const HostTensor& Detector::tensor(const std::string& name) const {
  auto it = tensors_.find(name);
  if (it == tensors_.end()) fail(-1, "missing tensor '%s'", name.c_str());
  return it->second;
}

void* Detector::alloc(size_t bytes) {
  bufs_.emplace_back(bytes);
  GTX_HIP(hipMemset(bufs_.back().p, 0, bufs_.back().bytes));
  return bufs_.back().p;
}

View Detector::new_view(int h, int w, int c) {
  View v;
  v.n = cfg_.max_batch;
  v.h = h;
  v.w = w;
  v.cstride = c;
  v.coff = 0;
  v.c = c;
  v.ptr = alloc((size_t)v.n * h * w * c * es_);
  return v;
}

namespace {
bool env_flag(const char* name, bool dflt) {
  const char* e = getenv(name);
  return (e && *e) ? e[0] != '0' : dflt;
}

// Packed weight images are a pure function of (tensor bytes, tile configuration). An engine builds several detectors from
// the same tensors (one per stream) and a run builds engines video after video: the image is made once per process and
// shared (packing YOLOv8s takes ~0.15 s of host time per detector, most of what creating one costs).
struct PackedWeights {
  std::vector<uint8_t> bytes;
  float acc_scale = 1.f;
  std::vector<float> source;      // the tensor the image was packed from: a hit is a hit only when these floats are the caller's
};
std::shared_ptr<const PackedWeights> packed_weights(const HostTensor& w, int cout, int cin, const ConvConfig& cfg,
                                                    const std::function<PackedWeights()>& make) {
  static std::mutex mu;
  static std::map<std::array<uint64_t, 4>, std::shared_ptr<const PackedWeights>> cache;
  uint64_t h = 1469598103934665603ull;                       // FNV-1a over the tensor's bytes, 8 at a time
  const uint64_t* q = reinterpret_cast<const uint64_t*>(w.data.data());
  for (size_t i = 0; i < w.data.size() / 2; ++i) h = (h ^ q[i]) * 1099511628211ull;
  if (w.data.size() & 1) h = (h ^ (uint64_t)__builtin_bit_cast(uint32_t, w.data.back())) * 1099511628211ull;
  const std::array<uint64_t, 4> key = {h, (uint64_t)w.data.size(), ((uint64_t)cout << 32) | (uint64_t)cin,
                                       ((uint64_t)cfg.dtype << 40) | ((uint64_t)cfg.ks << 32) | ((uint64_t)cfg.bn << 16) | ((uint64_t)cfg.kc << 4) | (uint64_t)cfg.variant};
  // The key's 64-bit FNV-1a is a filter, not an identity: a hit must also hold the same floats (a collision between two layers or
  // checkpoints of one shape would otherwise run the detector on another tensor's weights, silently). Bounded by bytes: the
  // images + sources of a YOLOv8x are ~1 GB; past 2 GB the oldest entries go.
  static std::deque<std::array<uint64_t, 4>> order;
  static size_t held = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end() && it->second->source.size() == w.data.size() &&
        memcmp(it->second->source.data(), w.data.data(), w.data.size() * sizeof(float)) == 0)
      return it->second;
  }
  PackedWeights fresh = make();
  fresh.source = w.data;
  auto made = std::make_shared<const PackedWeights>(std::move(fresh));
  const size_t cost = made->bytes.size() + made->source.size() * sizeof(float);
  std::lock_guard<std::mutex> lk(mu);
  auto old = cache.find(key);
  if (old != cache.end()) {                                    // same key, other floats: the newer tensor takes the slot
    held -= old->second->bytes.size() + old->second->source.size() * sizeof(float);
    cache.erase(old);
    order.erase(std::remove(order.begin(), order.end(), key), order.end());
  }
  while (!order.empty() && held + cost > ((size_t)2 << 30)) {
    auto victim = cache.find(order.front());
    if (victim != cache.end()) {
      held -= victim->second->bytes.size() + victim->second->source.size() * sizeof(float);
      cache.erase(victim);                                     // detectors that use the image keep it alive through their shared_ptr
    }
    order.pop_front();
  }
  held += cost;
  order.push_back(key);
  return cache.emplace(key, std::move(made)).first->second;
}

// OIHW -> OHWI
std::vector<float> to_ohwi(const HostTensor& t) {
  const int O = (int)t.shape[0], I = (int)t.shape[1], KH = (int)t.shape[2], KW = (int)t.shape[3];
  std::vector<float> r((size_t)O * I * KH * KW);
  parallel_for(O, [&](int o) {
    for (int i = 0; i < I; ++i)
      for (int y = 0; y < KH; ++y)
        for (int x = 0; x < KW; ++x)
          r[(((size_t)o * KH + y) * KW + x) * I + i] = t.data[(((size_t)o * I + i) * KH + y) * KW + x];
  });
  return r;
}
}  // namespace

// Emits one Conv op: weights "<name>.weight" (OIHW) / "<name>.bias" (optional).
View Detector::conv(const std::string& name, const View& x, int stride, bool act, const View* out_slice,
                    const View* residual, const View* up_src) {
  const HostTensor& w = tensor(name + ".weight");
  GTX_CHECK(w.shape.size() == 4 && w.shape[2] == w.shape[3], "%s: expected OIHW square kernel", name.c_str());
  const int cout = (int)w.shape[0], cin = (int)w.shape[1], ks = (int)w.shape[2];
  GTX_CHECK(cin == x.c, "%s: weight expects %d input channels, input view has %d", name.c_str(), cin, x.c);
  const int pad = ks / 2;
  const int ho = (x.h + 2 * pad - ks) / stride + 1, wo = (x.w + 2 * pad - ks) / stride + 1;
  View out = out_slice ? *out_slice : new_view(ho, wo, cout);
  GTX_CHECK(out.h == ho && out.w == wo && out.c == cout, "%s: output view mismatch", name.c_str());
  if (conv_dtype_ == DT_F32S) {     // pair format: whole 8-channel groups everywhere
    out.plain = plain_out_;
    GTX_CHECK(!x.plain && (!residual || !residual->plain) && (!up_src || !up_src->plain), "%s: a plain fp32 tensor cannot feed a split convolution", name.c_str());
    GTX_CHECK(x.cstride % 8 == 0 && x.coff % 8 == 0 && out.cstride % 8 == 0 && out.coff % 8 == 0 &&
                  (!residual || (residual->cstride % 8 == 0 && residual->coff % 8 == 0)),
              "%s: channel strides / offsets of the split-f16x3 path must be multiples of 8", name.c_str());
  }

  Op op;
  op.kind = Op::CONV;
  op.name = name;
  op.cfg = conv_pick_config(conv_dtype_, ks, stride, cin, cout, force_kc_, force_bn_, (long)x.n * ho * wo);
  const auto pw = packed_weights(w, cout, cin, op.cfg, [&] {
    PackedWeights r;
    const std::vector<float> ohwi = to_ohwi(w);
    r.bytes = pack_conv_weights(ohwi.data(), cout, cin, op.cfg, &r.acc_scale);
    return r;
  });
  const float acc_scale = pw->acc_scale;
  const std::vector<uint8_t>& packed = pw->bytes;
  void* dw = alloc(packed.size());
  GTX_HIP(hipMemcpy(dw, packed.data(), packed.size(), hipMemcpyHostToDevice));
  float* db = nullptr;
  if (has(name + ".bias")) {
    const HostTensor& b = tensor(name + ".bias");
    GTX_CHECK((int)b.data.size() == cout, "%s: bias size", name.c_str());
    db = (float*)alloc(((cout + 63) / 64 * 64) * sizeof(float));      // zero-filled up to a whole cout tile: the kernels load a tile's bias unconditionally
    GTX_HIP(hipMemcpy(db, b.data.data(), cout * sizeof(float), hipMemcpyHostToDevice));
  }
  ConvProblem& p = op.grp.p[0];
  p.in = x.ptr; p.out = out.ptr; p.wpack = dw; p.bias = db;
  p.res = residual ? residual->ptr : nullptr;
  p.N = x.n; p.H = x.h; p.W = x.w; p.Ho = ho; p.Wo = wo; p.Cin = cin; p.Cout = cout;
  p.in_cstride = x.cstride; p.in_coff = x.coff;
  p.out_cstride = out.cstride; p.out_coff = out.coff;
  p.res_cstride = residual ? residual->cstride : 0;
  p.res_coff = residual ? residual->coff : 0;
  p.act = act ? 1 : 0;
  p.acc_scale = acc_scale;
  p.out_plain = out.plain ? 1 : 0;
  p.sat_flag = conv_dtype_ == DT_F32S ? sat_dev_ : nullptr;
  if (up_src) {
    if (op.cfg.variant == 6) op.cfg.variant = 2;     // the second source is read by the 32x32x16 kernel only (same weight image)
    GTX_CHECK(ks == 1 && stride == 1 && op.cfg.variant == 2 && up_src->h * 2 == x.h && up_src->w * 2 == x.w && up_src->c < cin,
              "%s: upsampled source does not fit", name.c_str());
    p.in2 = up_src->ptr; p.in2_cstride = up_src->cstride; p.in2_coff = up_src->coff; p.c_split = up_src->c;
  }
  op.grp.count = 1;
  op.family = conv_kernel_name(op.cfg);
  ops_.push_back(op);
  layer_views_[name] = out;
  return out;
}

View Detector::c2f(const std::string& pfx, const View& x, bool shortcut, const View* out_slice, const View* up_src) {
  const HostTensor& w1 = tensor(pfx + ".cv1.conv.weight");
  const int c = (int)w1.shape[0] / 2;
  int n = 0;
  while (has(pfx + ".m." + std::to_string(n) + ".cv1.conv.weight")) ++n;
  View cat = new_view(x.h, x.w, (2 + n) * c);
  View first = cat.slice(0, 2 * c);
  conv(pfx + ".cv1.conv", x, 1, true, &first, nullptr, up_src);
  for (int k = 0; k < n; ++k) {
    // a buffer of its own per bottleneck: rows that no launch rewrites (plan_pad_skip) must keep ONE producer's values
    View tmp = new_view(x.h, x.w, c);
    const std::string m = pfx + ".m." + std::to_string(k);
    View src = cat.slice((1 + k) * c, c);
    View dst = cat.slice((2 + k) * c, c);
    conv(m + ".cv1.conv", src, 1, true, &tmp, nullptr);
    conv(m + ".cv2.conv", tmp, 1, true, &dst, shortcut ? &src : nullptr);
  }
  View out = conv(pfx + ".cv2.conv", cat, 1, true, out_slice, nullptr);
  layer_views_[pfx] = out;
  return out;
}

void Detector::build_graph() {
  const int H = lb_.net_h, W = lb_.net_w;
  img_ = new_view(H, W, 4);
  if (conv_dtype_ == DT_F32S) {
    sat_dev_ = (int*)alloc(sizeof(int));
    GTX_HIP(hipHostMalloc((void**)&h_sat_, sizeof(int)));
    *h_sat_ = 0;
  }

  // ---- layer 0: stem (dedicated 3-channel kernel) ----
  const HostTensor& w0 = tensor("model.0.conv.weight");
  GTX_CHECK(w0.shape.size() == 4 && w0.shape[1] == 3 && w0.shape[2] == 3, "model.0 must be a 3x3 conv on 3 channels");
  const int c0 = (int)w0.shape[0];
  View a0 = new_view(H / 2, W / 2, c0);
  {
    std::vector<float> w27((size_t)27 * c0);
    for (int o = 0; o < c0; ++o)
      for (int i = 0; i < 3; ++i)
        for (int y = 0; y < 3; ++y)
          for (int x = 0; x < 3; ++x)
            w27[(size_t)((y * 3 + x) * 3 + i) * c0 + o] = w0.data[(((size_t)o * 3 + i) * 3 + y) * 3 + x];
    float* dw = (float*)alloc(w27.size() * sizeof(float));
    GTX_HIP(hipMemcpy(dw, w27.data(), w27.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> b(c0, 0.f);
    if (has("model.0.conv.bias")) b = tensor("model.0.conv.bias").data;
    float* db = (float*)alloc((size_t)(c0 + 31) / 32 * 32 * sizeof(float));   // zero-filled up to whole 32-channel groups (the MFMA stems read a group's bias unconditionally)
    GTX_HIP(hipMemcpy(db, b.data(), c0 * sizeof(float), hipMemcpyHostToDevice));
    Op op;
    op.kind = Op::STEM;
    op.name = "model.0.conv";
    op.family = dtype_ == DT_F16 ? "stem_mfma_kernel" : (conv_dtype_ == DT_F32S ? "stem_split_kernel" : "stem_kernel");
    op.in = img_;
    op.out = a0;
    op.w27 = dw;
    op.bias = db;
    if (dtype_ == DT_F16) {
      const std::vector<uint16_t> pk = pack_stem_weights_f16(w27.data(), c0);
      void* dp = alloc(pk.size() * 2);
      GTX_HIP(hipMemcpy(dp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
      op.wpk = dp;
    } else if (conv_dtype_ == DT_F32S) {
      const std::vector<uint16_t> pk = pack_stem_weights_split(w27.data(), c0, &op.stem_scale);
      void* dp = alloc(pk.size() * 2);
      GTX_HIP(hipMemcpy(dp, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
      op.wpk = dp;
      if (c0 % 16 == 0) {                       // fuse_stem(): whole 16-channel K chunks of model.1
        const std::vector<uint16_t> fk = pack_front_weights_split(w27.data(), c0, &op.front_scale);
        void* fp = alloc(fk.size() * 2);
        GTX_HIP(hipMemcpy(fp, fk.data(), fk.size() * 2, hipMemcpyHostToDevice));
        op.front_wpk = fp;
      }
    }
    ops_.push_back(op);
    layer_views_["model.0.conv"] = a0;
  }

  auto cout_of = [&](const std::string& n) { return (int)tensor(n + ".weight").shape[0]; };

  // ---- backbone ----
  View a1 = conv("model.1.conv", a0, 2, true, nullptr, nullptr);
  View a2 = c2f("model.2", a1, true, nullptr);
  View a3 = conv("model.3.conv", a2, 2, true, nullptr, nullptr);
  // model.4 output feeds conv5 and Concat(14) = [up13, model.4]
  const int c4 = cout_of("model.4.cv2.conv"), c6 = cout_of("model.6.cv2.conv");
  const int c9 = cout_of("model.9.cv2.conv"), c12 = cout_of("model.12.cv2.conv");
  const int c16 = cout_of("model.16.conv"), c19 = cout_of("model.19.conv");
  View cat14 = new_view(H / 8, W / 8, c12 + c4);
  View s4 = cat14.slice(c12, c4);
  View a4 = c2f("model.4", a3, true, &s4);
  View a5 = conv("model.5.conv", a4, 2, true, nullptr, nullptr);
  View cat11 = new_view(H / 16, W / 16, c9 + c6);
  View s6 = cat11.slice(c9, c6);
  View a6 = c2f("model.6", a5, true, &s6);
  View a7 = conv("model.7.conv", a6, 2, true, nullptr, nullptr);
  View a8 = c2f("model.8", a7, true, nullptr);
  // ---- SPPF (model.9): cv1 -> 3 cascaded pools -> cv2; output lives in Concat(20) = [conv19, model.9]
  View cat20 = new_view(H / 32, W / 32, c19 + c9);
  View s9 = cat20.slice(c19, c9);
  {
    const int cm = cout_of("model.9.cv1.conv");
    View sp = new_view(a8.h, a8.w, 4 * cm);
    View sp0 = sp.slice(0, cm);
    conv("model.9.cv1.conv", a8, 1, true, &sp0, nullptr);
    Op op;
    op.kind = Op::POOL;
    op.name = "model.9.m";
    op.family = "sppf_pool_kernel";
    op.in = sp0;
    op.out = sp;
    ops_.push_back(op);
    conv("model.9.cv2.conv", sp, 1, true, &s9, nullptr);
    layer_views_["model.9"] = s9;
  }
  // ---- head ----
  auto upsample = [&](const std::string& name, const View& src, const View& dst) {
    Op op;
    op.kind = Op::UPSAMPLE;
    op.name = name;
    op.family = "upsample2x_kernel";
    op.in = src;
    op.out = dst;
    ops_.push_back(op);
  };
  // torch's Upsample + Concat in front of model.12 / model.15: the split-f16x3 path reads the low-resolution tensor in
  // place from the C2f's first 1x1 conv (ConvProblem::in2), the other arithmetics write the upsampled copy
  const bool fuse_up = conv_dtype_ == DT_F32S && c9 % 32 == 0 && c12 % 32 == 0;
  if (!fuse_up) upsample("model.10", s9, cat11.slice(0, c9));
  View cat17 = new_view(H / 16, W / 16, c16 + c12);
  View s12 = cat17.slice(c16, c12);
  c2f("model.12", cat11, false, &s12, fuse_up ? &s9 : nullptr);
  if (!fuse_up) upsample("model.13", s12, cat14.slice(0, c12));
  View a15 = c2f("model.15", cat14, false, nullptr, fuse_up ? &s12 : nullptr);
  View s16 = cat17.slice(0, c16);
  conv("model.16.conv", a15, 2, true, &s16, nullptr);
  View a18 = c2f("model.18", cat17, false, nullptr);
  View s19 = cat20.slice(0, c19);
  conv("model.19.conv", a18, 2, true, &s19, nullptr);
  View a21 = c2f("model.21", cat20, false, nullptr);

  // ---- Detect (model.22) ----
  // Stage 1 fuses the sibling convs cv2[l][0] and cv3[l][0] (same input) into one conv by
  // stacking their output channels; stage 2 runs cv2[l][1] and cv3[l][1] on channel slices.
  // The three levels go out as one grouped launch per stage. The final 1x1 convs are folded
  // into the decode kernels (the box one only runs for anchors that pass the score gate).
  const View lvl_in[3] = {a15, a18, a21};
  const float strides[3] = {8.f, 16.f, 32.f};
  std::vector<Op> st1, st2;
  head_ = HeadParams{};
  head_.n_levels = 3;
  head_.nc = cfg_.nc;
  head_.conf = cfg_.conf;
  head_.class_mask[0] = head_.class_mask[1] = cfg_.n_classes == 0 ? ~0ull : 0ull;
  for (int i = 0; i < cfg_.n_classes; ++i)
    if (cfg_.classes[i] >= 0 && cfg_.classes[i] < 128) head_.class_mask[cfg_.classes[i] >> 6] |= 1ull << (cfg_.classes[i] & 63);
  int anchor = 0;
  for (int l = 0; l < 3; ++l) {
    const std::string b2 = "model.22.cv2." + std::to_string(l), b3 = "model.22.cv3." + std::to_string(l);
    const HostTensor &w20 = tensor(b2 + ".0.conv.weight"), &w30 = tensor(b3 + ".0.conv.weight");
    const int cb = (int)w20.shape[0], cc = (int)w30.shape[0], cin = (int)w20.shape[1];
    GTX_CHECK(cin == lvl_in[l].c && (int)w30.shape[1] == cin, "Detect level %d input channels", l);
    // stacked stage-1 weights / bias
    HostTensor ws;
    ws.shape = {cb + cc, cin, 3, 3};
    ws.data = w20.data;
    ws.data.insert(ws.data.end(), w30.data.begin(), w30.data.end());
    tensors_["__head" + std::to_string(l) + ".s1.weight"] = ws;
    HostTensor bs;
    bs.shape = {cb + cc};
    bs.data = tensor(b2 + ".0.conv.bias").data;
    const auto& b30 = tensor(b3 + ".0.conv.bias").data;
    bs.data.insert(bs.data.end(), b30.begin(), b30.end());
    tensors_["__head" + std::to_string(l) + ".s1.bias"] = bs;
    const size_t mark = ops_.size();
    // The three levels run as grouped launches: one K chunk and one cout tile for all members of a stage. Widths that
    // are multiples of 16 only (yolov8 n / m / x) take the 16-channel chunk; a stage with a member whose Cout is not a
    // multiple of 64 takes the 32-cout tile.
    bool k32 = true;
    for (int q = 0; q < 3; ++q) k32 = k32 && lvl_in[q].c % 32 == 0;
    force_kc_ = conv_dtype_ == DT_F16 ? (k32 ? 32 : 16) : 0;
    force_bn_ = (cb + cc) % 64 == 0 ? 64 : 32;
    View h1 = conv("__head" + std::to_string(l) + ".s1", lvl_in[l], 1, true, nullptr, nullptr);
    View h2 = new_view(h1.h, h1.w, cb + cc);
    View h1b = h1.slice(0, cb), h1c = h1.slice(cb, cc), h2b = h2.slice(0, cb), h2c = h2.slice(cb, cc);
    force_kc_ = conv_dtype_ == DT_F16 ? ((cb % 32 == 0 && cc % 32 == 0) ? 32 : 16) : 0;
    force_bn_ = (cb % 64 == 0 && cc % 64 == 0) ? 64 : 32;
    plain_out_ = true;            // the decode kernels read these two as plain fp32
    conv(b2 + ".1.conv", h1b, 1, true, &h2b, nullptr);
    conv(b3 + ".1.conv", h1c, 1, true, &h2c, nullptr);
    plain_out_ = false;
    h2.plain = conv_dtype_ == DT_F32S;
    force_kc_ = force_bn_ = 0;
    // move the three freshly built single-problem ops into the grouped stage ops: one grouped launch per stage and kernel
    // configuration (the levels of a stage share a launch when they share the kernel; the fp32 default path runs its deep
    // levels -- Cin >= 256 -- on the Winograd kernel and the 240 x 240 level on the direct one: two launches for a stage)
    GTX_CHECK(ops_.size() == mark + 3, "internal: head op count");
    Op o1 = ops_[mark], o2 = ops_[mark + 1], o3 = ops_[mark + 2];
    ops_.resize(mark);
    // Sparse box branch (head_sparse.hip): the box half of stage 1 (cout tile 0 of the stacked image) and cv2[l][1] are evaluated
    // at the candidate anchors only, after the score gate; stage 1 keeps its class half (cout tiles 1..), same packed image,
    // same scale. Needs the 16x16x32 kernel's image (32-channel chunks, one 64-cout box tile); decided for all levels at once.
    if (l == 0) {
      sparse_on_ = conv_dtype_ == DT_F32S && env_flag("GTX_SPARSE_BOX", true);
      sparse_ = SparseBox{};
      dense_box_ops_.clear();
    }
    sparse_on_ = sparse_on_ && cb == 64 && o1.cfg.variant == 5 && o1.cfg.bn == 64 && o2.cfg.variant == 5 && cin % 32 == 0;
    head_ops_[l][0] = o1; head_ops_[l][1] = o2; head_ops_[l][2] = o3;

    HeadLevel& L = head_.lv[l];
    L.feat = h2.ptr; L.h = h2.h; L.w = h2.w; L.cstride = h2.cstride; L.cb = cb; L.cc = cc;
    L.stride = strides[l];
    L.anchor_begin = anchor;
    anchor += h2.h * h2.w;
    auto upload = [&](const std::vector<float>& v) {
      float* d = (float*)alloc(v.size() * sizeof(float));
      GTX_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
      return d;
    };
    const HostTensor &wb = tensor(b2 + ".2.weight"), &wc = tensor(b3 + ".2.weight");
    GTX_CHECK(wb.shape[0] == 64 && (int)wb.shape[1] == cb, "Detect box head must have 4*16 outputs");
    GTX_CHECK(cb <= 128 && cc % 8 == 0, "Detect head widths cb=%d cc=%d are outside what the decode kernels support", cb, cc);
    GTX_CHECK(l == 0 || cb == head_.lv[0].cb, "Detect box branch width differs between levels");
    GTX_CHECK((int)wc.shape[0] == cfg_.nc && (int)wc.shape[1] == cc, "Detect cls head has %d outputs, nc=%d", (int)wc.shape[0], cfg_.nc);
    {
      std::vector<float> wbt((size_t)cb * 64);  // [cb][64]: the decode wave reads one output per lane
      for (int o = 0; o < 64; ++o)
        for (int k = 0; k < cb; ++k) wbt[(size_t)k * 64 + o] = wb.data[(size_t)o * cb + k];
      L.wb = upload(wbt);
    }
    L.bb = upload(tensor(b2 + ".2.bias").data);
    L.wc = upload(wc.data); L.bc = upload(tensor(b3 + ".2.bias").data);
    layer_views_["model.22.feat" + std::to_string(l)] = h2;
  }
  head_.n_anchors = anchor;
  {
    auto same = [](const ConvConfig& a, const ConvConfig& b) {
      return a.dtype == b.dtype && a.ks == b.ks && a.stride == b.stride && a.bn == b.bn && a.kc == b.kc &&
             a.variant == b.variant && a.th == b.th && a.tw == b.tw;
    };
    auto add = [&](std::vector<Op>& stage, const char* name, const Op& o) {
      for (Op& g : stage)
        if (same(g.cfg, o.cfg) && g.grp.count < kMaxGroup) { g.grp.p[g.grp.count++] = o.grp.p[0]; return; }
      Op g;
      g.kind = Op::CONV;
      g.name = stage.empty() ? std::string(name) : std::string(name) + "." + std::to_string(stage.size());
      g.cfg = o.cfg;
      g.grp.p[g.grp.count++] = o.grp.p[0];
      stage.push_back(g);
    };
    std::vector<Op> d1, d2;                          // the dense box layers, kept for the debug read-backs and the overflow case
    for (int l = 0; l < 3; ++l) {
      Op o1 = head_ops_[l][0];
      const Op &o2 = head_ops_[l][1], &o3 = head_ops_[l][2];
      if (sparse_on_) {
        const ConvProblem full = o1.grp.p[0];
        const int cb = head_.lv[l].cb, cc = head_.lv[l].cc;
        const size_t tile_bytes = (size_t)(full.Cin / o1.cfg.kc) * 9 * 64 * 128;   // one 64-cout tile of the packed image
        SparseBoxLevel& S = sparse_.lv[l];
        S.in = full.in; S.H = full.H; S.W = full.W; S.cstride = full.in_cstride; S.coff = full.in_coff; S.cin = full.Cin;
        S.w1 = full.wpack; S.b1 = full.bias; S.sc1 = full.acc_scale;
        S.w2 = o2.grp.p[0].wpack; S.b2 = o2.grp.p[0].bias; S.sc2 = o2.grp.p[0].acc_scale;
        S.anchor_begin = head_.lv[l].anchor_begin;
        S.wb = head_.lv[l].wb; S.bb = head_.lv[l].bb; S.stride = head_.lv[l].stride;
        GTX_CHECK(full.bias && o2.grp.p[0].bias && o2.grp.p[0].Cin == 64 && o2.grp.p[0].Cout == 64, "sparse box branch: unexpected Detect box layers");
        Op box1 = o1;                                // the box tile alone
        box1.grp.p[0].Cout = cb;
        add(d1, "model.22.box1", box1);
        add(d2, "model.22.box2", o2);
        ConvProblem& p = o1.grp.p[0];                // the class tiles alone
        p.wpack = static_cast<const char*>(full.wpack) + tile_bytes;
        p.bias = full.bias + 64;
        p.Cout = cc;
        p.out_coff = full.out_coff + cb;
        add(st1, "model.22.stage1", o1);
        add(st2, "model.22.stage2", o3);
      } else {
        add(st1, "model.22.stage1", o1);
        add(st2, "model.22.stage2", o2);
        add(st2, "model.22.stage2", o3);
      }
    }
    if (sparse_on_) {
      sparse_.n_levels = 3;
      for (std::vector<Op>* stage : {&d1, &d2})
        for (Op& g : *stage) { g.family = conv_kernel_name(g.cfg); dense_box_ops_.push_back(g); }
    }
  }
  feat_levels_ = FeatLevels{};
  if (cfg_.obj_feats) {                           // `with_reid: true, model: auto`: the Detect layer's inputs, read after NMS
    feat_levels_.n_levels = 3;
    feat_levels_.dim = std::min(lvl_in[0].c, std::min(lvl_in[1].c, lvl_in[2].c));
    for (int l = 0; l < 3; ++l) {
      feat_levels_.feat[l] = lvl_in[l].ptr; feat_levels_.h[l] = lvl_in[l].h; feat_levels_.w[l] = lvl_in[l].w;
      feat_levels_.cstride[l] = lvl_in[l].cstride; feat_levels_.coff[l] = lvl_in[l].coff; feat_levels_.c[l] = lvl_in[l].c;
      feat_levels_.anchor_begin[l] = head_.lv[l].anchor_begin;
      GTX_CHECK(lvl_in[l].c % feat_levels_.dim == 0, "obj_feats: Detect input %d has %d channels, not a multiple of %d", l, lvl_in[l].c, feat_levels_.dim);
    }
  }
  for (std::vector<Op>* stage : {&st1, &st2})
    for (Op& g : *stage) {
      g.family = conv_kernel_name(g.cfg);
      ops_.push_back(g);
    }
}

// model.1.conv (3x3 stride 2, all of its output channels in one cout tile) and model.2.cv1.conv (the 1x1 that is its only
// consumer) become one launch: ConvProblem::post_w. The 3x3 layer's output is never written; the launch writes the 1x1
// layer's. GTX_FUSE_FRONT=0 keeps the two launches. YOLOv8 n and s qualify (32 / 64 channels); the wider scales do not.
void Detector::fuse_front() {
  const char* e = getenv("GTX_FUSE_FRONT");
  if (conv_dtype_ != DT_F32S || (e && e[0] == '0')) return;
  for (size_t i = 0; i + 1 < ops_.size(); ++i) {
    if (ops_[i].name != "model.1.conv" || ops_[i + 1].name != "model.2.cv1.conv") continue;
    const Op &a = ops_[i], &b = ops_[i + 1];
    if (a.kind != Op::CONV || b.kind != Op::CONV || a.grp.count != 1 || b.grp.count != 1) return;
    const ConvProblem &pa = a.grp.p[0], &pb = b.grp.p[0];
    const bool ok = a.cfg.variant == 2 && a.cfg.ks == 3 && a.cfg.stride == 2 && pa.Cout == a.cfg.bn && !pa.res &&
                    (b.cfg.variant == 2 || b.cfg.variant == 6) && b.cfg.ks == 1 && b.cfg.kc == 32 && b.cfg.bn == pb.Cout && pb.Cin == pa.Cout && pb.Cout == pa.Cout &&
                    pb.in == pa.out && pb.in_cstride == pa.out_cstride && pb.in_coff == pa.out_coff && !pb.res && !pb.in2;
    if (!ok) return;
    unfused_ = {a, b};
    Op f = a;
    f.name = "model.1.conv+model.2.cv1.conv";
    ConvProblem& p = f.grp.p[0];
    p.post_w = pb.wpack; p.post_bias = pb.bias; p.post_scale = pb.acc_scale; p.post_act = pb.act;
    p.out = pb.out; p.out_cstride = pb.out_cstride; p.out_coff = pb.out_coff; p.out_plain = pb.out_plain;
    ops_[i] = f;
    ops_.erase(ops_.begin() + (long)i + 1);
    return;
  }
}

// model.0.conv (the stem) moves into the launch of its only consumer, model.1.conv (already carrying model.2.cv1 when
// fuse_front() applied): ConvProblem::front_img. The stem's output -- the largest tensor of the network, 236 MB per two
// 1920 x 1920 frames -- is neither written nor read back; the workgroup recomputes the one-pixel halo of its patch (9.6 %).
// Built for model.1 in one cout tile and at most two 16-channel K chunks: YOLOv8 n (16 -> 32) and s (32 -> 64).
// GTX_FUSE_STEM=0 keeps the stem's own launch.
void Detector::fuse_stem() {
  const char* e = getenv("GTX_FUSE_STEM");
  if (conv_dtype_ != DT_F32S || (e && e[0] == '0') || ops_.size() < 2) return;
  const Op& st = ops_[0];
  Op& cv = ops_[1];
  if (st.kind != Op::STEM || cv.kind != Op::CONV || cv.grp.count != 1 || !st.front_wpk) return;
  ConvProblem& p = cv.grp.p[0];
  const bool ok = cv.cfg.variant == 2 && cv.cfg.ks == 3 && cv.cfg.stride == 2 && cv.cfg.kc == 16 && cv.cfg.th == 8 && ((cv.cfg.bn == 32 && p.Cin == 16) || (cv.cfg.bn == 64 && p.Cin == 32)) &&
                  p.Cout <= cv.cfg.bn && !p.res && p.in == st.out.ptr && p.in_cstride == st.out.c && p.in_coff == 0 && p.Cin == st.out.c &&
                  st.in.h == 2 * st.out.h && st.in.w == 2 * st.out.w && p.H == st.out.h && p.W == st.out.w;
  if (!ok) return;
  unfused_.insert(unfused_.begin(), st);
  p.front_img = st.in.ptr;
  p.front_w = st.front_wpk;
  p.front_bias = st.bias;
  p.front_scale = st.front_scale;
  p.front_h = st.in.h;
  p.front_w_px = st.in.w;
  cv.name = "model.0.conv+" + cv.name;
  cv.family = "conv_front_split_kernel";
  ops_.erase(ops_.begin());
}

// The stem's output and model.1's are written by no launch of the fused forward pass (354 MB per two 1920 x 1920 frames and
// detector): their buffers are given back here. layer_output() of one of them re-creates them and runs the stand-alone
// launches (unfused_). Until then every reference to them holds a token that is no device address.
namespace {
void swap_ptr(Op& o, const void* from, void* to) {
  if (o.in.ptr == from) o.in.ptr = to;
  if (o.out.ptr == from) o.out.ptr = to;
  for (int i = 0; i < o.grp.count; ++i) {
    ConvProblem& p = o.grp.p[i];
    if (p.in == from) p.in = to;
    if (p.out == from) p.out = to;
    if (p.res == from) p.res = to;
    if (p.in2 == from) p.in2 = to;
  }
}
}  // namespace

void Detector::release_hidden_layers() {
  std::vector<void*> ptrs;
  for (size_t i = 0; i + 1 < unfused_.size() || (i < unfused_.size() && unfused_[i].kind == Op::STEM); ++i) {
    const Op& o = unfused_[i];                       // every stand-alone op but the last conv writes a hidden tensor
    void* out = o.kind == Op::STEM ? o.out.ptr : (o.grp.count == 1 ? o.grp.p[0].out : nullptr);
    if (out) ptrs.push_back(out);
  }
  for (void* ptr : ptrs) {
    bool live = false;                               // still written or read by a launch of the forward pass?
    for (const Op& o : ops_) {
      if (o.kind != Op::CONV && (o.in.ptr == ptr || o.out.ptr == ptr)) live = true;
      for (int i = 0; i < o.grp.count; ++i) {
        const ConvProblem& q = o.grp.p[i];
        const bool reads_in = !q.front_img;          // a front stage computes its input patch from the image instead of loading it
        if ((reads_in && q.in == ptr) || q.out == ptr || q.res == ptr || q.in2 == ptr) live = true;
      }
    }
    if (live) continue;
    for (size_t b = 0; b < bufs_.size(); ++b) {
      if (bufs_[b].p != ptr) continue;
      void* token = reinterpret_cast<void*>(static_cast<uintptr_t>(16 * (hidden_.size() + 1)));
      hidden_.push_back({token, bufs_[b].bytes, nullptr});
      bufs_.erase(bufs_.begin() + (long)b);
      for (Op& o : unfused_) swap_ptr(o, ptr, token);
      for (auto& kv : layer_views_)
        if (kv.second.ptr == ptr) kv.second.ptr = token;
      break;
    }
  }
}

void Detector::materialize_hidden_layers() {
  for (Hidden& h : hidden_) {
    if (h.real) continue;
    h.real = alloc(h.bytes);
    for (Op& o : unfused_) swap_ptr(o, h.token, h.real);
    for (auto& kv : layer_views_)
      if (kv.second.ptr == h.token) kv.second.ptr = h.real;
  }
}

void Detector::set_batch(int nb) {
  if (nb == cur_nb_) return;
  for (Op& op : ops_) {
    if (op.kind != Op::CONV) continue;
    for (int i = 0; i < op.grp.count; ++i) {
      op.grp.p[i].N = nb;
      op.grp.p[i].ty_first = pad_skip_on_ ? op.ty_first[i] : 0;
      op.grp.p[i].ty_count = pad_skip_on_ ? op.ty_count[i] : 0;
    }
    conv_group_finalize(op.grp, op.cfg);
    op.flops = 0;
    op.bytes = 0;
    for (int i = 0; i < op.grp.count; ++i) {
      const ConvProblem& p = op.grp.p[i];
      // the share of the output rows this launch computes (letterbox-padding rows are skipped: plan_pad_skip)
      const double part = std::min(1.0, (double)p.tiles_y * op.cfg.th / p.Ho);
      op.flops += part * conv_flops(p, op.cfg.ks);
      op.bytes += part * ((double)p.N * p.H * p.W * (p.Cin - 0.75 * p.c_split) + (double)p.N * p.Ho * p.Wo * p.Cout) * es_ +
                  (double)p.Cout * p.Cin * op.cfg.ks * op.cfg.ks * es_;
      if (p.post_w) {                     // the fused 1x1 layer: its FLOPs and weights; its output replaces the 3x3 layer's (same size)
        op.flops += part * 2.0 * p.N * p.Ho * p.Wo * (double)p.Cout * p.Cout;
        op.bytes += (double)p.Cout * p.Cout * es_;
      }
      if (p.front_img) {                  // the fused stem: its FLOPs; RGB0 bytes are read instead of the stem's output
        op.flops += part * 2.0 * p.N * p.H * p.W * (double)p.Cin * 27;
        op.bytes += part * ((double)p.N * p.front_h * p.front_w_px * 4 - (double)p.N * p.H * p.W * p.Cin * es_);
      }
    }
  }
  for (Op& op : ops_) {
    if (op.kind == Op::STEM) {
      op.flops = 2.0 * nb * op.out.h * op.out.w * op.out.c * 27;
      op.bytes = (double)nb * ((double)op.in.h * op.in.w * 4 + (double)op.out.h * op.out.w * op.out.c * es_);   // RGB0 bytes in
    } else if (op.kind == Op::POOL) {
      op.flops = 0;
      op.bytes = (double)nb * op.in.h * op.in.w * op.in.c * 4 * es_;
    } else if (op.kind == Op::UPSAMPLE) {
      op.flops = 0;
      op.bytes = (double)nb * op.in.h * op.in.w * op.in.c * 5 * es_;
    }
  }
  cur_nb_ = nb;
}

// Rows of every activation tensor that can depend on the frame. The letterbox puts the resized frame in rows [top, top + new_h)
// of the network input and a constant colour everywhere else (ultralytics LetterBox, default.yaml `rect: false`: 420 + 420 of
// 1920 rows for a 16:9 frame); a convolution's output row outside the reach of those rows sees the same inputs for every
// frame, so its value is a constant of the checkpoint. Those rows are computed once (prime_pad_skip, at finalize) and the
// launches of every later pass cover the other tile rows only -- the buffers keep the constants, the results are the full
// launches' bit for bit. The interval grows with every 3x3 layer and covers the whole map from the SPPF on; the gain is in the
// backbone's large maps. GTX_PAD_SKIP=0: off.
void Detector::plan_pad_skip() {
  pad_skip_rows_ = pad_skip_total_ = 0;
  struct Rows { int lo, hi; };
  std::map<const void*, Rows> dep;                   // buffer -> frame-dependent rows; a buffer that is not here depends on the frame everywhere
  std::set<const void*> full;                        // ... and these stay that way whatever is written to them later
  // A region two different launches write (a scratch buffer reused by two layers) cannot keep either's constants: whoever
  // writes it computes every row. Writers per buffer as channel ranges.
  struct Writer { int c0, c1; Op* op; int member; };
  std::map<const void*, std::vector<Writer>> writers;
  auto second_writer = [&](const void* p, int c0, int c1, Op* op, int member) {
    bool hit = false;
    for (Writer& w : writers[p])
      if (c0 < w.c1 && w.c0 < c1) {
        hit = true;
        if (w.op->ty_count[w.member] > 0) {            // the earlier writer had been given a row range: take it back
          const ConvProblem& q = w.op->grp.p[w.member];
          pad_skip_rows_ -= (q.Ho + w.op->cfg.th - 1) / w.op->cfg.th - w.op->ty_count[w.member];
          w.op->ty_first[w.member] = w.op->ty_count[w.member] = 0;
        }
      }
    writers[p].push_back(Writer{c0, c1, op, member});
    return hit;
  };
  dep[img_.ptr] = Rows{lb_.top, lb_.top + lb_.new_h};
  auto through = [](Rows r, int k, int s, int h_out) {   // rows of a k x k / stride s / pad k/2 layer's output that see input rows [lo, hi)
    const int p = k / 2;
    const int num = r.lo + p - k + 1;                      // y >= num / s
    const int lo = num <= 0 ? 0 : (num + s - 1) / s;
    const int hi = (r.hi - 1 + p) / s + 1;
    return Rows{lo, std::min(h_out, hi)};
  };
  auto unite = [](Rows a, Rows b) { return Rows{std::min(a.lo, b.lo), std::max(a.hi, b.hi)}; };
  auto get = [&](const void* p, Rows& r) {
    auto it = dep.find(p);
    if (it == dep.end()) return false;
    r = it->second;
    return true;
  };
  auto put = [&](const void* p, bool known, Rows r) {
    if (!known) { dep.erase(p); full.insert(p); return; }
    if (full.count(p)) return;
    auto it = dep.find(p);
    if (it == dep.end()) dep[p] = r; else it->second = unite(it->second, r);   // another slice of a concat buffer may reach further
  };
  for (Op& op : ops_) {
    if (op.kind == Op::STEM) {
      Rows in{0, 0};
      const bool known = get(op.in.ptr, in);
      put(op.out.ptr, known, through(in, 3, 2, op.out.h));
      continue;
    }
    if (op.kind != Op::CONV) { put(op.out.ptr, false, Rows{0, 0}); continue; }   // pools / upsampling: deep in the network
    for (int i = 0; i < op.grp.count; ++i) {
      const ConvProblem& p = op.grp.p[i];
      op.ty_first[i] = op.ty_count[i] = 0;
      Rows in{0, p.H};
      bool known;
      if (p.front_img) {
        Rows img{0, 0};
        known = get(p.front_img, img);
        if (known) in = through(img, 3, 2, p.H);
      } else {
        known = get(p.in, in) && p.c_split == 0;       // a second, upsampled source: neck layers, all rows
      }
      Rows out = through(in, op.cfg.ks, op.cfg.stride, p.Ho);
      if (known && p.res) {
        Rows r{0, 0};
        known = get(p.res, r);
        if (known) out = unite(out, r);
      }
      const int tiles = (p.Ho + op.cfg.th - 1) / op.cfg.th;
      pad_skip_total_ += tiles;
      if (second_writer(p.out, p.out_coff, p.out_coff + p.Cout, &op, i)) known = false;
      put(p.out, known, out);
      if (!known) continue;
      const int t0 = out.lo / op.cfg.th, t1 = std::min(tiles, (out.hi + op.cfg.th - 1) / op.cfg.th);
      if (t1 - t0 < tiles && t1 > t0) {
        op.ty_first[i] = t0;
        op.ty_count[i] = t1 - t0;
        pad_skip_rows_ += tiles - (t1 - t0);
      }
    }
  }
}

// One full pass over every batch slot on a blank frame: the rows the later launches skip now hold their constants.
void Detector::prime_pad_skip() {
  if (pad_skip_rows_ == 0) return;
  const int N = cfg_.max_batch;
  hipStream_t s = ctx_->stream;
  DevBuf blank;
  blank.alloc((size_t)N * cfg_.frame_h * cfg_.frame_w * 3);
  GTX_HIP(hipMemsetAsync(blank.p, 0, blank.bytes, s));
  pad_skip_on_ = false;
  cur_nb_ = 0;
  set_batch(N);
  launch_preprocess(dtype_, (const uint8_t*)blank.p, N, lb_, img_.ptr, nullptr, gray_h_, gray_w_, s);
  for (const Op& op : ops_) run_op(op, N, s);
  int sat = 0;
  if (sat_dev_) GTX_HIP(hipMemcpyAsync(&sat, sat_dev_, sizeof(int), hipMemcpyDeviceToHost, s));
  GTX_HIP(hipStreamSynchronize(s));
  cur_nb_ = 0;
  if (sat_dev_) GTX_HIP(hipMemsetAsync(sat_dev_, 0, sizeof(int), s));
  if (sat) {
    // A constant of the padding rows lies beyond fp16's range: it was clamped on its way into the pair format. With the rows
    // left out of the later launches no pass would raise the flag for them again, so the skipping stays off for this detector:
    // every pass computes (and checks) every row, and the first one falls back to the exact-fp32 kernels (collect()).
    pad_skip_on_ = false;
    pad_skip_rows_ = 0;
    for (Op& op : ops_)
      for (int i = 0; i < kMaxGroup; ++i) op.ty_first[i] = op.ty_count[i] = 0;
    return;
  }
  pad_skip_on_ = true;
}

void Detector::finalize() {
  GTX_CHECK(!finalized_, "finalize called twice");
  GTX_HIP(hipSetDevice(ctx_->device));
  build_graph();
  fuse_front();
  fuse_stem();
  release_hidden_layers();
  const int N = cfg_.max_batch;
  gray_h_ = cfg_.frame_h / 2;
  gray_w_ = cfg_.frame_w / 2;
  gray_.alloc((size_t)kGrayRing * N * gray_h_ * gray_w_);
  // NMS workspace. Candidate capacity = every anchor; sort/NMS capacity = ultralytics max_nms.
  nms_ = NmsBuffers{};
  nms_.cap = head_.n_anchors;
  nms_.nms_cap = 30016;  // >= max_nms (30000), multiple of 64
  nms_.max_det = cfg_.max_det;
  nms_.count = (int*)alloc(sizeof(int) * N);
  nms_.cand_score = (float*)alloc(sizeof(float) * N * nms_.cap);
  nms_.cand_anchor = (int*)alloc(sizeof(int) * N * nms_.cap);
  nms_.cand_cls = (int*)alloc(sizeof(int) * N * nms_.cap);
  nms_.cand_box = (float*)alloc(sizeof(float) * 4 * N * nms_.cap);
  nms_.sorted_n = (int*)alloc(sizeof(int) * N);
  nms_.s_box = (float*)alloc(sizeof(float) * 4 * N * nms_.nms_cap);
  nms_.s_score = (float*)alloc(sizeof(float) * N * nms_.nms_cap);
  nms_.s_cls = (int*)alloc(sizeof(int) * N * nms_.nms_cap);
  nms_.mask = (unsigned long long*)alloc(sizeof(unsigned long long) * N * (size_t)nms_.nms_cap * (nms_.nms_cap / 64));
  nms_.out_n = (int*)alloc(sizeof(int) * N);
  nms_.out_rows = (float*)alloc(sizeof(float) * 6 * N * cfg_.max_det);
  nms_.s_anchor = nms_.out_anchor = nullptr;
  if (cfg_.obj_feats) {
    nms_.s_anchor = (int*)alloc(sizeof(int) * N * nms_.nms_cap);
    nms_.out_anchor = (int*)alloc(sizeof(int) * N * cfg_.max_det);
    d_feats_ = (float*)alloc(sizeof(float) * N * cfg_.max_det * feat_levels_.dim);
    GTX_HIP(hipHostMalloc((void**)&h_feats_, sizeof(float) * N * cfg_.max_det * feat_levels_.dim));
  }
  if (sparse_on_) {
    sparse_.cap = kSparseCap;
    sparse_.sat_flag = sat_dev_;
    nms_.lvl_cap = kSparseCap;
    nms_.lvl_count = (int*)alloc(sizeof(int) * N * kMaxLevels);
    nms_.lvl_list = (int*)alloc(sizeof(int) * N * kMaxLevels * kSparseCap);
  }
  GTX_HIP(hipHostMalloc((void**)&h_count_, sizeof(int) * N));
  GTX_HIP(hipHostMalloc((void**)&h_out_n_, sizeof(int) * N));
  GTX_HIP(hipHostMalloc((void**)&h_out_rows_, sizeof(float) * 6 * N * cfg_.max_det));
  if (conv_dtype_ != DT_F32S) tensors_.clear();  // host copies are no longer needed (the split path keeps them for fall_back_to_exact)
  if (env_flag("GTX_PAD_SKIP", true)) {
    plan_pad_skip();
    prime_pad_skip();
  }
  set_batch(1);
  GTX_HIP(hipStreamSynchronize(ctx_->stream));
  finalized_ = true;
}

void Detector::run_op(const Op& op, int nb, hipStream_t s) {
  switch (op.kind) {
    case Op::CONV:
      conv_launch(op.grp, op.cfg, s);
      break;
    case Op::STEM:
      launch_stem(dtype_ == DT_F32 ? conv_dtype_ : dtype_, op.in.ptr, nb, op.in.h, op.in.w, op.w27, op.bias, op.wpk, op.out.c, op.out.ptr, op.out.h,
                  op.out.w, s, op.stem_scale);
      break;
    case Op::POOL: launch_sppf_pool(conv_dtype_ == DT_F32S ? DT_F32S : dtype_, op.out.ptr, nb, op.in.h, op.in.w, op.in.c, s); break;
    case Op::UPSAMPLE:
      launch_upsample2x(dtype_, op.in.ptr, nb, op.in.h, op.in.w, op.in.c, op.in.cstride, op.in.coff, op.out.ptr,
                        op.out.cstride, op.out.coff, s);
      break;
  }
}

void Detector::run_forward(int nb, hipStream_t s, bool traced) {
  if (!traced) {
    // The forward graph is static for a given batch size (every launch has fixed arguments), so it can be
    // captured once into a hipGraph and replayed as one submission (GTX_GRAPH=1). Measured on MI355X: no
    // gain (1292 vs 1274 frames/s end to end, 823 vs 832 detector-only at batch 1) -- the ~5 us between
    // dependent launches is GPU-side dispatch, not host submission -- so plain launches stay the default.
    static const bool use_graph = [] { const char* e = getenv("GTX_GRAPH"); return e && e[0] == '1'; }();
    if (use_graph) {
      if (graph_nb_ != nb) {
        if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
        hipGraph_t g = nullptr;
        GTX_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (const Op& op : ops_) run_op(op, nb, s);
        GTX_HIP(hipStreamEndCapture(s, &g));
        GTX_HIP(hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        graph_nb_ = nb;
      }
      GTX_HIP(hipGraphLaunch(graph_exec_, s));
      return;
    }
    for (const Op& op : ops_) run_op(op, nb, s);
    return;
  }
  for (size_t i = 0; i < ops_.size(); ++i) {
    GTX_HIP(hipEventRecord(trace_ev_[i], s));
    run_op(ops_[i], nb, s);
  }
  GTX_HIP(hipEventRecord(trace_ev_[ops_.size()], s));
}

void Detector::set_trace(int every_n) {
  if (exact_) return exact_->set_trace(every_n);
  GTX_CHECK(finalized_ && every_n >= 0, "set_trace: detector not finalized or bad period");
  GTX_CHECK(!in_flight_, "set_trace while a batch is in flight");
  trace_every_ = every_n;
  trace_count_ = 0;
  if (every_n > 0 && trace_ev_.empty()) {
    trace_ev_.resize(ops_.size() + 1);
    for (auto& e : trace_ev_) GTX_HIP(hipEventCreate(&e));
  }
  trace_ms_.assign(ops_.size(), 0.0);
  trace_n_.assign(ops_.size(), 0);
  trace_flops_.assign(ops_.size(), 0.0);
  trace_bytes_.assign(ops_.size(), 0.0);
}

void Detector::trace_report(std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms,
                            std::vector<double>& flops, std::vector<double>& bytes) {
  if (exact_) return exact_->trace_report(names, launches, ms, flops, bytes);
  std::map<std::string, size_t> idx;
  for (size_t i = 0; i < ops_.size() && i < trace_n_.size(); ++i) {
    if (trace_n_[i] == 0) continue;
    auto it = idx.find(ops_[i].family);
    size_t k;
    if (it == idx.end()) {
      k = names.size();
      idx[ops_[i].family] = k;
      names.push_back(ops_[i].family);
      launches.push_back(0); ms.push_back(0.f); flops.push_back(0.0); bytes.push_back(0.0);
    } else {
      k = it->second;
    }
    launches[k] += trace_n_[i];
    ms[k] += (float)trace_ms_[i];
    flops[k] += trace_flops_[i];
    bytes[k] += trace_bytes_[i];
  }
  trace_ms_.assign(ops_.size(), 0.0);
  trace_n_.assign(ops_.size(), 0);
  trace_flops_.assign(ops_.size(), 0.0);
  trace_bytes_.assign(ops_.size(), 0.0);
}

void Detector::run_post(int nb, hipStream_t s) {
  if (sparse_on_) {                                // score gate -> the box branch at the candidates -> their boxes
    launch_head_gate(dtype_, head_, nb, nms_, s);
    launch_head_sparse_box(sparse_, nb, nms_, s);    // the box branch at the candidates and their boxes
    dense_head_valid_ = false;
  } else {
    launch_head_candidates(dtype_, head_, nb, nms_, s);
  }
  // the candidate counts come back with the results: collect() runs what this pass leaves out for a batch that needs it (the
  // general NMS kernels for > 4096 candidates in an image, the dense box layers for > kSparseCap)
  GTX_HIP(hipMemcpyAsync(h_count_, nms_.count, sizeof(int) * nb, hipMemcpyDeviceToHost, s));
  launch_nms(nms_, nb, cfg_.iou, cfg_.agnostic_nms != 0, 30000, lb_, s, 1);
  GTX_HIP(hipMemcpyAsync(h_out_n_, nms_.out_n, sizeof(int) * nb, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipMemcpyAsync(h_out_rows_, nms_.out_rows, sizeof(float) * 6 * nb * cfg_.max_det, hipMemcpyDeviceToHost, s));
  if (sat_dev_) GTX_HIP(hipMemcpyAsync(h_sat_, sat_dev_, sizeof(int), hipMemcpyDeviceToHost, s));
  if (cfg_.obj_feats) {
    launch_obj_feats(conv_dtype_, feat_levels_, nb, nms_, d_feats_, s);
    GTX_HIP(hipMemcpyAsync(h_feats_, d_feats_, sizeof(float) * nb * cfg_.max_det * feat_levels_.dim, hipMemcpyDeviceToHost, s));
  }
}

// The dense box layers of the Detect head on the activations of the pass that ran last (sparse mode leaves them out of the
// forward): for the debug read-backs and for a batch with more candidates than the sparse buffer holds.
void Detector::run_dense_box(hipStream_t s) {
  if (!sparse_on_ || dense_head_valid_) return;
  GTX_CHECK(cur_nb_ > 0, "no forward pass has run yet");
  for (Op o : dense_box_ops_) {
    for (int i = 0; i < o.grp.count; ++i) o.grp.p[i].N = cur_nb_;
    conv_group_finalize(o.grp, o.cfg);
    run_op(o, cur_nb_, s);
  }
  dense_head_valid_ = true;
}

void Detector::features(int b, float* out, int cap, int* n, int* dim) const {
  if (exact_) return exact_->features(b, out, cap, n, dim);
  GTX_CHECK(cfg_.obj_feats && h_feats_, "features: the detector was created without gtx_det_config.obj_feats");
  GTX_CHECK(b >= 0 && b < cfg_.max_batch, "features: image %d of %d", b, cfg_.max_batch);
  const int cnt = std::min(b < (int)c_feat_n_.size() ? c_feat_n_[b] : 0, cap);
  if (n) *n = cnt;
  if (dim) *dim = feat_levels_.dim;
  if (out && cnt > 0) memcpy(out, c_feats_.data() + (size_t)b * cfg_.max_det * feat_levels_.dim, sizeof(float) * cnt * feat_levels_.dim);
}

// A split-f16x3 pass clamped an activation: from here on this object is a shell around an exact-fp32 detector built from
// the same tensors on the same context. The split graph's device memory (activations, packed weights) is given back.
void Detector::fall_back_to_exact() {
  gtx_det_config c = cfg_;
  c.fp32_split = 0;
  std::unique_ptr<Detector> d(new Detector(ctx_, c));
  for (const auto& kv : tensors_) d->set_tensor(kv.first, kv.second.data.data(), (int)kv.second.shape.size(), kv.second.shape.data());
  d->finalize();
  if (trace_every_ > 0) d->set_trace(trace_every_);
  GTX_HIP(hipStreamSynchronize(ctx_->stream));
  if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
  ops_.clear();
  unfused_.clear();
  layer_views_.clear();
  bufs_.clear();
  tensors_.clear();
  exact_ = std::move(d);
}

bool Detector::saturated(bool clear) {
  const bool r = sat_seen_;
  if (clear) {
    sat_seen_ = false;
    if (sat_dev_ && !exact_) {
      GTX_HIP(hipSetDevice(ctx_->device));
      GTX_HIP(hipMemsetAsync(sat_dev_, 0, sizeof(int), ctx_->stream));
    }
  }
  return r;
}

// Asynchronous half: enqueue preprocess -> forward -> decode/NMS -> D2H of the result rows on the
// context's stream and return. Results are picked up by collect(). The gray image of this batch
// goes to the next slot of a 16-deep ring so that consumers on other streams (stabilizers) can
// still read the images of the four batches before the newest collected one.
void Detector::submit_dev(const void* frames, int nb, int h, int w) {
  if (exact_) return exact_->submit_dev(frames, nb, h, w);
  GTX_CHECK(finalized_, "detector not finalized");
  GTX_CHECK(!in_flight_, "submit while a batch is in flight: call collect first");
  GTX_CHECK(nb >= 1 && nb <= cfg_.max_batch, "batch %d outside [1,%d]", nb, cfg_.max_batch);
  GTX_CHECK(h == cfg_.frame_h && w == cfg_.frame_w, "frame is %dx%d, detector was created for %dx%d", w, h, cfg_.frame_w, cfg_.frame_h);
  GTX_HIP(hipSetDevice(ctx_->device));
  hipStream_t s = ctx_->stream;
  set_batch(nb);
  cur_frames_ = frames;
  gray_slot_ = (gray_slot_ + 1) % kGrayRing;
  uint8_t* gray = gray_.as<uint8_t>() + (size_t)gray_slot_ * cfg_.max_batch * gray_h_ * gray_w_;
  GTX_HIP(hipEventRecord(ev_[0], s));
  launch_preprocess(dtype_, (const uint8_t*)frames, nb, lb_, img_.ptr, gray, gray_h_, gray_w_, s);
  GTX_HIP(hipEventRecord(ev_[1], s));
  flight_traced_ = trace_every_ > 0 && (trace_count_++ % trace_every_) == 0;
  run_forward(nb, s, flight_traced_);
  GTX_HIP(hipEventRecord(ev_[2], s));
  run_post(nb, s);
  GTX_HIP(hipEventRecord(ev_[3], s));
  in_flight_ = true;
  flight_nb_ = nb;
}

void Detector::collect(int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) {
  if (exact_) return exact_->collect(n_out, xyxy, conf, cls, speed_ms);
  GTX_CHECK(in_flight_, "collect without a submitted batch");
  GTX_HIP(hipSetDevice(ctx_->device));
  GTX_HIP(hipEventSynchronize(ev_[3]));
  in_flight_ = false;
  collected_gray_slot_ = gray_slot_;
  if (h_sat_ && *h_sat_) {
    sat_seen_ = true;
    static const bool fallback = [] { const char* e = getenv("GTX_SAT_FALLBACK"); return !(e && e[0] == '0'); }();
    if (fallback && conv_dtype_ == DT_F32S && !tensors_.empty()) {
      // this batch again, at fp32's range: the frames are still where the caller put them (one batch in flight per detector)
      flight_traced_ = false;
      fall_back_to_exact();
      return exact_->detect_dev(cur_frames_, flight_nb_, cfg_.frame_h, cfg_.frame_w, n_out, xyxy, conf, cls, speed_ms);
    }
  }
  if (flight_traced_) {
    for (size_t i = 0; i < ops_.size(); ++i) {
      float t = 0.f;
      GTX_HIP(hipEventElapsedTime(&t, trace_ev_[i], trace_ev_[i + 1]));
      trace_ms_[i] += t;
      trace_n_[i] += 1;
      trace_flops_[i] += ops_[i].flops;          // of THIS pass's batch size (set_batch ran in submit_dev)
      trace_bytes_[i] += ops_[i].bytes;
    }
    flight_traced_ = false;
  }
  {                                                  // what the pass left out for the common case
    bool over = false, big = false;
    for (int b = 0; b < flight_nb_; ++b) {
      over = over || (sparse_on_ && h_count_[b] > kSparseCap);                 // more candidates than the sparse buffer holds
      big = big || !nms_small_covers(std::min(h_count_[b], nms_.cap), nms_.max_det);   // ... than the single-workgroup NMS takes
    }
    if (over || big) {
      hipStream_t s = ctx_->stream;
      NmsBuffers dense = nms_;
      if (over) {                                    // the dense box layers, then every candidate's box again
        run_dense_box(s);
        launch_head_boxes(dtype_, head_, flight_nb_, dense, s);
        ++sparse_overflows_;
      }
      launch_nms(dense, flight_nb_, cfg_.iou, cfg_.agnostic_nms != 0, 30000, lb_, s, over ? 0 : 2);
      GTX_HIP(hipMemcpyAsync(h_out_n_, nms_.out_n, sizeof(int) * flight_nb_, hipMemcpyDeviceToHost, s));
      GTX_HIP(hipMemcpyAsync(h_out_rows_, nms_.out_rows, sizeof(float) * 6 * flight_nb_ * cfg_.max_det, hipMemcpyDeviceToHost, s));
      if (cfg_.obj_feats) {
        launch_obj_feats(conv_dtype_, feat_levels_, flight_nb_, nms_, d_feats_, s);
        GTX_HIP(hipMemcpyAsync(h_feats_, d_feats_, sizeof(float) * flight_nb_ * cfg_.max_det * feat_levels_.dim, hipMemcpyDeviceToHost, s));
      }
      GTX_HIP(hipEventRecord(ev_[3], s));            // the postprocess figure (ev_[2] -> ev_[3]) now includes the re-run
      GTX_HIP(hipStreamSynchronize(s));
    }
  }
  const int nb = flight_nb_;
  if (cfg_.obj_feats) {                              // out of the pinned buffer before the next pass of this detector lands in it
    c_feat_n_.assign(cfg_.max_batch, 0);
    c_feats_.resize((size_t)cfg_.max_batch * cfg_.max_det * feat_levels_.dim);
    for (int b = 0; b < nb; ++b) {
      c_feat_n_[b] = h_out_n_[b];
      const size_t o = (size_t)b * cfg_.max_det * feat_levels_.dim;
      memcpy(c_feats_.data() + o, h_feats_ + o, sizeof(float) * h_out_n_[b] * feat_levels_.dim);
    }
  }
  for (int b = 0; b < nb; ++b) {
    const int n = h_out_n_[b];
    n_out[b] = n;
    const float* rows = h_out_rows_ + (size_t)b * cfg_.max_det * 6;
    for (int i = 0; i < n; ++i) {
      float* bx = xyxy + ((size_t)b * cfg_.max_det + i) * 4;
      bx[0] = rows[i * 6 + 0]; bx[1] = rows[i * 6 + 1]; bx[2] = rows[i * 6 + 2]; bx[3] = rows[i * 6 + 3];
      conf[(size_t)b * cfg_.max_det + i] = rows[i * 6 + 4];
      cls[(size_t)b * cfg_.max_det + i] = (int)rows[i * 6 + 5];
    }
  }
  if (speed_ms) {
    for (int i = 0; i < 3; ++i) GTX_HIP(hipEventElapsedTime(&speed_ms[i], ev_[i], ev_[i + 1]));
  }
}

void Detector::detect_dev(const void* frames, int nb, int h, int w, int* n_out, float* xyxy, float* conf,
                          int* cls, float speed_ms[3]) {
  submit_dev(frames, nb, h, w);
  collect(n_out, xyxy, conf, cls, speed_ms);
}

void Detector::detect_host(const uint8_t* frame, int h, int w, int* n_out, float* xyxy, float* conf, int* cls,
                           float speed_ms[3]) {
  GTX_CHECK(finalized_, "detector not finalized");
  GTX_HIP(hipSetDevice(ctx_->device));
  const size_t bytes = (size_t)h * w * 3;
  if (frame_stage_.bytes < bytes) frame_stage_.alloc(bytes);
  GTX_HIP(hipEventRecord(ev_up_[0], ctx_->stream));
  GTX_HIP(hipMemcpyAsync(frame_stage_.p, frame, bytes, hipMemcpyHostToDevice, ctx_->stream));
  GTX_HIP(hipEventRecord(ev_up_[1], ctx_->stream));
  detect_dev(frame_stage_.p, 1, h, w, n_out, xyxy, conf, cls, speed_ms);
  if (speed_ms) {
    float up_ms = 0.f;
    GTX_HIP(hipEventElapsedTime(&up_ms, ev_up_[0], ev_up_[1]));
    speed_ms[0] += up_ms;  // the host->device copy is part of "preprocess"
  }
}

const void* Detector::gray(int b, int* gh, int* gw) const {
  if (exact_) return exact_->gray(b, gh, gw);
  if (gh) *gh = gray_h_;
  if (gw) *gw = gray_w_;
  if (b < 0 || b >= cfg_.max_batch) return nullptr;
  // the image of the most recently *collected* batch (a newer batch may already be in flight)
  return gray_.as<uint8_t>() + ((size_t)collected_gray_slot_ * cfg_.max_batch + b) * gray_h_ * gray_w_;
}

void Detector::raw_output(int b, float* out, int* n_anchors, bool logits) {
  if (exact_) return exact_->raw_output(b, out, n_anchors, logits);
  GTX_CHECK(finalized_ && cur_nb_ > 0 && b >= 0 && b < cur_nb_, "raw_output: no forward pass for slot %d", b);
  const size_t per = (size_t)head_.n_anchors * (4 + head_.nc);
  if (raw_.bytes < per * cur_nb_ * sizeof(float)) raw_.alloc(per * cur_nb_ * sizeof(float));
  GTX_CHECK(!in_flight_, "raw_output while a batch is in flight: call collect first");
  run_dense_box(ctx_->stream);                     // the full decode reads every anchor's box features
  launch_head_raw(dtype_, head_, cur_nb_, raw_.as<float>(), logits, ctx_->stream);
  GTX_HIP(hipMemcpyAsync(out, raw_.as<float>() + per * b, per * sizeof(float), hipMemcpyDeviceToHost, ctx_->stream));
  GTX_HIP(hipStreamSynchronize(ctx_->stream));
  if (n_anchors) *n_anchors = head_.n_anchors;
}

void Detector::layer_output(int b, const std::string& layer, float* out, int* h, int* w, int* c) {
  if (exact_) return exact_->layer_output(b, layer, out, h, w, c);
  // model.0 / model.1 of a fused front launch are RECOMPUTED below by their stand-alone launches (same products, another
  // summation order): what comes back is not what the network consumed. Not while a pass is in flight: the launches
  // would queue behind it and overwrite the buffers it shares with them.
  GTX_CHECK(!(out && in_flight_), "layer_output while a batch is in flight: call collect first");
  auto it = layer_views_.find(layer);
  if (it == layer_views_.end()) fail(-1, "unknown layer '%s'", layer.c_str());
  const View& v = it->second;
  if (out && sparse_on_ && cur_nb_ > 0 && (layer.rfind("model.22.", 0) == 0 || layer.rfind("__head", 0) == 0)) {
    GTX_HIP(hipSetDevice(ctx_->device));
    run_dense_box(ctx_->stream);                   // the head's box layers are not part of the forward in sparse mode
    GTX_HIP(hipStreamSynchronize(ctx_->stream));
  }
  if (out) {
    // an intermediate the fused launches no longer write (the stem's output, model.1's): run the stand-alone launches up to
    // it now -- the network input of the last pass is still in HBM. The last entry of unfused_ is a layer the fused launch
    // does write.
    int k = -1;
    for (size_t i = 0; i < unfused_.size(); ++i) {
      const bool still_written = i + 1 == unfused_.size() && unfused_[i].kind != Op::STEM;   // the fused launch's own output layer
      if (!still_written && unfused_[i].name == layer) k = (int)i;
    }
    if (k >= 0) GTX_CHECK(cur_nb_ > 0, "layer_output('%s'): no forward pass has run yet", layer.c_str());
    if (k >= 0 && cur_nb_ > 0) {
      GTX_HIP(hipSetDevice(ctx_->device));
      materialize_hidden_layers();
      for (int j = 0; j <= k; ++j) {
        Op o = unfused_[(size_t)j];
        if (o.kind == Op::CONV) {
          for (int i = 0; i < o.grp.count; ++i) o.grp.p[i].N = cur_nb_;
          conv_group_finalize(o.grp, o.cfg);
        }
        run_op(o, cur_nb_, ctx_->stream);
      }
      GTX_HIP(hipStreamSynchronize(ctx_->stream));
    }
  }
  if (h) *h = v.h;
  if (w) *w = v.w;
  if (c) *c = v.c;
  if (!out) return;
  GTX_CHECK(b >= 0 && b < cfg_.max_batch, "bad batch slot");
  GTX_CHECK(reinterpret_cast<uintptr_t>(v.ptr) > 4096, "layer_output('%s'): the layer's buffer was not re-created", layer.c_str());
  const size_t px = (size_t)v.h * v.w;
  std::vector<uint8_t> host(px * v.cstride * es_);
  GTX_HIP(hipMemcpy(host.data(), (const uint8_t*)v.ptr + (size_t)b * px * v.cstride * es_, host.size(), hipMemcpyDeviceToHost));
  for (size_t p = 0; p < px; ++p)
    for (int k = 0; k < v.c; ++k) {
      const size_t src = p * v.cstride + v.coff + k;
      float f;
      if (conv_dtype_ == DT_F32S && !v.plain) {
        f = pair_element(host.data(), src);
      } else if (dtype_ == DT_F16) {
        _Float16 hv;
        memcpy(&hv, host.data() + src * 2, 2);
        f = (float)hv;
      } else {
        memcpy(&f, host.data() + src * 4, 4);
      }
      out[p * v.c + k] = f;
    }
}

void Detector::profile(int nb, int iters, std::vector<std::string>& names, std::vector<int>& launches,
                       std::vector<float>& ms, std::vector<double>& flops, std::vector<double>& bytes) {
  if (exact_) return exact_->profile(nb, iters, names, launches, ms, flops, bytes);
  GTX_CHECK(finalized_, "detector not finalized");
  GTX_CHECK(nb >= 1 && nb <= cfg_.max_batch && iters >= 1, "bad profile arguments");
  hipStream_t s = ctx_->stream;
  set_batch(nb);
  std::vector<hipEvent_t> ev(ops_.size() + 1);
  for (auto& e : ev) GTX_HIP(hipEventCreate(&e));
  std::map<std::string, size_t> idx;
  const bool per_op = std::getenv("GTX_PROFILE_PER_OP") != nullptr;   // one line per launch (its module path) instead of per kernel family
  auto slot = [&](const std::string& fam) {
    auto it = idx.find(fam);
    if (it != idx.end()) return it->second;
    idx[fam] = names.size();
    names.push_back(fam);
    launches.push_back(0);
    ms.push_back(0.f);
    flops.push_back(0.0);
    bytes.push_back(0.0);
    return names.size() - 1;
  };
  for (int it = 0; it < iters; ++it) {
    for (size_t i = 0; i < ops_.size(); ++i) {
      GTX_HIP(hipEventRecord(ev[i], s));
      run_op(ops_[i], nb, s);
    }
    GTX_HIP(hipEventRecord(ev[ops_.size()], s));
    GTX_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < ops_.size(); ++i) {
      float t = 0.f;
      GTX_HIP(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      const size_t k = slot(per_op ? (i < 10 ? "0" : "") + std::to_string(i) + " " + ops_[i].name : ops_[i].family);
      launches[k] += 1;
      ms[k] += t;
      flops[k] += ops_[i].flops;
      bytes[k] += ops_[i].bytes;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
}

}  // namespace gtx
