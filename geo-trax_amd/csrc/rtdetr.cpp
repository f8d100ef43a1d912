// RT-DETR graph builder + executor. Layer topology follows ultralytics' cfg/models/rt-detr/rtdetr-l.yaml (backbone 0-9:
// HGStem, HGBlock x 6 with DWConv downsampling; head 10-27: AIFI on P5, CCFM with RepC3; 28: RTDETRDecoder); channel widths,
// class count, decoder depth and block lengths are read off the tensor shapes.
#include "rtdetr.hpp"
#include "split_format.hpp"

#include <cmath>

namespace gtx {

namespace {
bool env_on(const char* name, bool dflt) {
  const char* e = getenv(name);
  return (e && *e) ? e[0] != '0' : dflt;
}
}  // namespace

RtDetr::RtDetr(gtx_ctx* ctx, const gtx_det_config& cfg) : ctx_(ctx), cfg_(cfg) {
  GTX_CHECK(cfg.imgsz > 0 && cfg.imgsz % 32 == 0, "imgsz must be a positive multiple of 32 (got %d)", cfg.imgsz);
  GTX_CHECK(cfg.max_det > 0 && cfg.nc > 0 && cfg.nc <= 128, "max_det must be positive and nc in [1, 128] (got %d, %d)", cfg.max_det, cfg.nc);
  GTX_CHECK(cfg.frame_h > 0 && cfg.frame_w > 0, "frame size must be given");
  GTX_CHECK(!cfg.obj_feats, "RT-DETR: appearance vectors (obj_feats) are not implemented");
  if (cfg_.max_batch < 1) cfg_.max_batch = 1;
  // half: fp16 maps and weights on the fp16 MFMA convolutions (fp32 accumulate); the token side (AIFI, the decoder's queries) stays fp32
  fmt_ = cfg.half ? DT_F16 : (cfg.fp32_split ? DT_F32S : DT_F32);
  // RTDETRPredictor.pre_transform: LetterBox(imgsz, auto=False, scale_fill=True) -- the frame is stretched to the square, no padding
  lb_ = Letterbox{};
  lb_.src_h = cfg.frame_h; lb_.src_w = cfg.frame_w;
  lb_.net_h = lb_.net_w = lb_.new_h = lb_.new_w = cfg.imgsz;
  lb_.top = lb_.left = 0;
  lb_.gain = 1.0;
  GTX_HIP(hipSetDevice(ctx->device));
  for (auto& e : ev_) GTX_HIP(hipEventCreateWithFlags(&e, wait_event_flags(true)));
  for (auto& e : ev_up_) GTX_HIP(hipEventCreate(&e));
}

RtDetr::~RtDetr() {
  if (h_out_n_) (void)hipHostFree(h_out_n_);
  if (h_out_rows_) (void)hipHostFree(h_out_rows_);
  if (h_sat_) (void)hipHostFree(h_sat_);
  for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
  for (auto& e : ev_up_) if (e) (void)hipEventDestroy(e);
  for (auto& e : trace_ev_) if (e) (void)hipEventDestroy(e);
}

void RtDetr::set_tensor(const std::string& name, const float* data, int ndim, const int64_t* shape) {
  GTX_CHECK(!finalized_, "set_tensor after finalize");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  tensors_[name] = std::move(t);
}

const HostTensor& RtDetr::tensor(const std::string& name) const {
  auto it = tensors_.find(name);
  if (it == tensors_.end()) fail(-1, "missing tensor '%s'", name.c_str());
  return it->second;
}

void* RtDetr::alloc(size_t bytes) {
  bufs_.emplace_back(bytes);
  GTX_HIP(hipMemset(bufs_.back().p, 0, bufs_.back().bytes));
  return bufs_.back().p;
}

float* RtDetr::upload(const std::vector<float>& v) {
  float* d = (float*)alloc(std::max<size_t>(v.size(), 1) * sizeof(float));
  if (!v.empty()) GTX_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

RtDetr::View RtDetr::new_view(int h, int w, int c, bool plain) {
  View v;
  v.n = cfg_.max_batch; v.h = h; v.w = w; v.cstride = c; v.coff = 0; v.c = c; v.plain = plain;
  v.ptr = alloc((size_t)v.n * h * w * c * 4);
  return v;
}

float* RtDetr::new_tokens(int rows_per_image, int ld, const std::string& name) {
  View v = new_view(1, rows_per_image, ld, true);
  if (!name.empty()) layer_views_[name] = v;
  return (float*)v.ptr;
}

// ---------------------------------------------------------------------------- graph pieces
RtDetr::View RtDetr::conv_raw(const std::string& name, const std::vector<float>& w, int cout, int cin, int ks, const float* bias_host, const View& x,
                              int stride, int act, const View* out_slice, const View* residual, bool plain_out) {
  GTX_CHECK(cin == x.c, "%s: weight expects %d input channels, input view has %d", name.c_str(), cin, x.c);
  GTX_CHECK((int)w.size() == cout * cin * ks * ks, "%s: weight size", name.c_str());
  const int pad = ks / 2;
  const int ho = (x.h + 2 * pad - ks) / stride + 1, wo = (x.w + 2 * pad - ks) / stride + 1;
  View out = out_slice ? *out_slice : new_view(ho, wo, cout, plain_out);
  GTX_CHECK(out.h == ho && out.w == wo && out.c == cout, "%s: output view mismatch", name.c_str());
  if (fmt_ == DT_F16) out.plain = false;                       // fp16 maps everywhere, the score maps included (rt_topk reads them as such)
  GTX_CHECK(!x.plain || fmt_ == DT_F32, "%s: a plain fp32 tensor cannot feed this convolution", name.c_str());
  if (fmt_ == DT_F32S)
    GTX_CHECK(x.cstride % 8 == 0 && x.coff % 8 == 0 && out.cstride % 8 == 0 && out.coff % 8 == 0 && (!residual || (residual->cstride % 8 == 0 && residual->coff % 8 == 0)),
              "%s: channel strides / offsets of the split-f16x3 path must be multiples of 8", name.c_str());
  Op op;
  op.kind = Op::CONV;
  op.name = name;
  op.cfg = conv_pick_config(fmt_, ks, stride, cin, cout);
  GTX_CHECK(op.cfg.variant != 3 && op.cfg.variant != 4, "RT-DETR does not run on the Winograd kernels (unset GTX_WINO)");
  std::vector<float> ohwi((size_t)cout * cin * ks * ks);       // OIHW -> OHWI
  parallel_for(cout, [&](int o) {
    for (int i = 0; i < cin; ++i)
      for (int t = 0; t < ks * ks; ++t) ohwi[((size_t)o * ks * ks + t) * cin + i] = w[((size_t)o * cin + i) * ks * ks + t];
  });
  float acc_scale = 1.f;
  const std::vector<uint8_t> packed = pack_conv_weights(ohwi.data(), cout, cin, op.cfg, &acc_scale);
  void* dw = alloc(packed.size());
  GTX_HIP(hipMemcpy(dw, packed.data(), packed.size(), hipMemcpyHostToDevice));
  float* db = (float*)alloc(((cout + 63) / 64 * 64) * sizeof(float));   // zero-filled up to a whole cout tile
  if (bias_host) GTX_HIP(hipMemcpy(db, bias_host, cout * sizeof(float), hipMemcpyHostToDevice));
  ConvProblem& p = op.grp.p[0];
  p.in = x.ptr; p.out = out.ptr; p.wpack = dw; p.bias = db;
  p.res = residual ? residual->ptr : nullptr;
  p.N = x.n; p.H = x.h; p.W = x.w; p.Ho = ho; p.Wo = wo; p.Cin = cin; p.Cout = cout;
  p.in_cstride = x.cstride; p.in_coff = x.coff;
  p.out_cstride = out.cstride; p.out_coff = out.coff;
  p.res_cstride = residual ? residual->cstride : 0;
  p.res_coff = residual ? residual->coff : 0;
  p.act = act;
  p.acc_scale = acc_scale;
  p.out_plain = (fmt_ == DT_F32S && out.plain) ? 1 : 0;
  p.sat_flag = fmt_ == DT_F32S ? sat_dev_ : nullptr;
  op.grp.count = 1;
  op.family = conv_kernel_name(op.cfg);
  ops_.push_back(op);
  layer_views_[name] = out;
  return out;
}

RtDetr::View RtDetr::conv(const std::string& name, const View& x, int stride, int act, const View* out_slice, const View* residual) {
  const HostTensor& w = tensor(name + ".weight");
  GTX_CHECK(w.shape.size() == 4 && w.shape[2] == w.shape[3], "%s: expected OIHW square kernel", name.c_str());
  const float* b = has(name + ".bias") ? tensor(name + ".bias").data.data() : nullptr;
  return conv_raw(name, w.data, (int)w.shape[0], (int)w.shape[1], (int)w.shape[2], b, x, stride, act, out_slice, residual);
}

// HGStem.stem2a / stem2b: a 2x2 convolution on F.pad(x, [0, 1, 0, 1]) is the 3x3 convolution (pad 1) whose kernel holds the
// four taps at rows / columns 1..2 and zeros in row 0 and column 0: the zero taps meet the left / top padding, the others
// the right / bottom one. Runs on the MFMA convolution as it stands (2.25 x the products of a 27 KFLOP-per-pixel layer).
RtDetr::View RtDetr::conv2x2(const std::string& name, const View& x, const View* out_slice) {
  const HostTensor& w = tensor(name + ".weight");
  GTX_CHECK(w.shape.size() == 4 && w.shape[2] == 2 && w.shape[3] == 2, "%s: expected a 2x2 kernel", name.c_str());
  const int cout = (int)w.shape[0], cin = (int)w.shape[1];
  std::vector<float> w3((size_t)cout * cin * 9, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int i = 0; i < cin; ++i)
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) w3[(((size_t)o * cin + i) * 3 + 1 + dy) * 3 + 1 + dx] = w.data[(((size_t)o * cin + i) * 2 + dy) * 2 + dx];
  const float* b = has(name + ".bias") ? tensor(name + ".bias").data.data() : nullptr;
  return conv_raw(name, w3, cout, cin, 3, b, x, 1, 2, out_slice, nullptr);
}

void RtDetr::dwconv(const std::string& name, const View& x, const View& out, int stride, int act) {
  const HostTensor& w = tensor(name + ".weight");
  GTX_CHECK(w.shape.size() == 4 && w.shape[1] == 1 && w.shape[2] == w.shape[3] && (int)w.shape[0] == x.c && out.c == x.c, "%s: expected a depthwise kernel on %d channels", name.c_str(), x.c);
  const int k = (int)w.shape[2], C = x.c;
  std::vector<float> wt((size_t)k * k * C);
  for (int c = 0; c < C; ++c)
    for (int t = 0; t < k * k; ++t) wt[(size_t)t * C + c] = w.data[(size_t)c * k * k + t];
  Op op;
  op.kind = Op::DWCONV;
  op.name = name;
  const bool tiled = stride == 1 && C % 32 == 0;               // launch_rt_dwconv's rule (GTX_RT_DW_TILE=0 aside)
  op.family = std::string(tiled ? "rt_dwconv_tile_kernel<" : "rt_dwconv_kernel<") + (k == 3 ? "3>" : "5>");
  op.a = x.map(); op.b = out.map();
  op.w = upload(wt);
  op.bias = has(name + ".bias") ? upload(tensor(name + ".bias").data) : nullptr;
  op.k = k; op.stride = stride; op.act = act;
  op.flops = 2.0 * out.h * out.w * C * k * k;
  op.bytes = ((double)x.h * x.w + (double)out.h * out.w) * C * 4;
  ops_.push_back(op);
  layer_views_[name] = out;
}

void RtDetr::upsample(const std::string& name, const View& src, const View& dst) {
  Op op;
  op.kind = Op::UPSAMPLE;
  op.name = name;
  op.family = "rt_upsample2x_kernel";
  op.a = src.map(); op.b = dst.map();
  op.bytes = (double)src.h * src.w * src.c * 4 * 5;
  ops_.push_back(op);
}

// The HGBlock's concat buffer [x | m.0 .. m.(n-1)]: allocated before x's producer so that x is written in place.
RtDetr::View RtDetr::hg_cat(const std::string& pfx, int h, int w, int* c1) {
  const bool light = has(pfx + ".m.0.conv1.conv.weight");
  const HostTensor& w0 = tensor(pfx + (light ? ".m.0.conv1.conv.weight" : ".m.0.conv.weight"));
  const int cm = (int)w0.shape[0];
  *c1 = (int)w0.shape[1];
  int n = 0;
  while (has(pfx + ".m." + std::to_string(n) + (light ? ".conv1.conv.weight" : ".conv.weight"))) ++n;
  return new_view(h, w, *c1 + n * cm);
}

RtDetr::View RtDetr::hgblock(const std::string& pfx, const View& cat, int c1, bool shortcut, const View* out_slice) {
  const bool light = has(pfx + ".m.0.conv1.conv.weight");
  const int cm = (int)tensor(pfx + (light ? ".m.0.conv1.conv.weight" : ".m.0.conv.weight")).shape[0];
  const int n = (cat.c - c1) / cm;
  const View x = cat.slice(0, c1);
  for (int i = 0; i < n; ++i) {
    const View src = i == 0 ? x : cat.slice(c1 + (i - 1) * cm, cm);
    const View dst = cat.slice(c1 + i * cm, cm);
    const std::string m = pfx + ".m." + std::to_string(i);
    if (light) {                                              // LightConv: 1x1 without activation, depthwise k x k + ReLU
      View tmp = new_view(cat.h, cat.w, cm);
      conv(m + ".conv1.conv", src, 1, 0, &tmp);
      dwconv(m + ".conv2.conv", tmp, dst, 1, 2);
    } else {
      conv(m + ".conv", src, 1, 2, &dst);
    }
  }
  View sc = conv(pfx + ".sc.conv", cat, 1, 2);
  const int c2 = (int)tensor(pfx + ".ec.conv.weight").shape[0];
  const bool add = shortcut && c1 == c2;
  View out = conv(pfx + ".ec.conv", sc, 1, 2, out_slice, add ? &x : nullptr);   // ReLU first, then + x
  layer_views_[pfx] = out;
  return out;
}

RtDetr::View RtDetr::repc3(const std::string& pfx, const View& x) {
  View a = conv(pfx + ".cv1.conv", x, 1, 1);
  View side = conv(pfx + ".cv2.conv", x, 1, 1);
  int n = 0;
  while (has(pfx + ".m." + std::to_string(n) + ".conv.weight")) ++n;
  GTX_CHECK(n >= 1, "%s: no RepConv blocks (fused `.m.i.conv` tensors expected)", pfx.c_str());
  for (int i = 0; i < n; ++i) a = conv(pfx + ".m." + std::to_string(i) + ".conv", a, 1, 1, nullptr, i == n - 1 ? &side : nullptr);   // m(cv1(x)) + cv2(x)
  if (has(pfx + ".cv3.conv.weight")) a = conv(pfx + ".cv3.conv", a, 1, 1);
  layer_views_[pfx] = a;
  return a;
}

// y = act(x (+ x2) . w^T + bias) (+ res) on token rows; w [nout][k] is zero-padded to multiples of 16 in both dims.
float* RtDetr::linear(const std::string& name, const std::vector<float>& w, const std::vector<float>& bias, int nout, int k, const float* x, int ldx,
                      const float* x2, int rows, int act, const float* res, int ldr, float* y, int ldy, const std::string& out_name) {
  GTX_CHECK((int)w.size() == nout * k && (bias.empty() || (int)bias.size() == nout), "%s: linear weight / bias size", name.c_str());
  const int np = (nout + 15) / 16 * 16, kp = (k + 15) / 16 * 16;
  std::vector<float> wp((size_t)np * kp, 0.f), bp(np, 0.f);
  for (int o = 0; o < nout; ++o) {
    memcpy(&wp[(size_t)o * kp], &w[(size_t)o * k], sizeof(float) * k);
    if (!bias.empty()) bp[o] = bias[o];
  }
  GTX_CHECK(kp <= ldx, "%s: input rows hold %d values, the padded K is %d", name.c_str(), ldx, kp);
  if (!y) { y = new_tokens(rows, np, out_name); ldy = np; }
  Op op;
  op.kind = Op::LINEAR;
  op.name = name;
  op.family = "rt_linear_kernel";
  RtLinear& L = op.lin;
  L.x = x; L.ldx = ldx; L.x2 = x2; L.ldx2 = ldx; L.x2_cols = np;
  L.w = upload(wp); L.bias = upload(bp);
  L.res = res; L.ldr = ldr;
  L.y = y; L.ldy = ldy;
  L.M = rows; L.K = kp; L.Nout = np; L.act = act;
  op.flops = 2.0 * rows * (double)k * nout;
  op.bytes = ((double)rows * (k + nout) + (double)k * nout) * 4;
  ops_.push_back(op);
  return y;
}

float* RtDetr::layernorm_tokens(const std::string& name, const float* x, int rows, int C, const std::string& out_name, const View* map_out) {
  Op op;
  op.kind = Op::LAYERNORM;
  op.name = name;
  op.family = "rt_layernorm_kernel";
  op.r_in = RtRows{const_cast<float*>(x), C, 0, DT_F32};
  float* y = nullptr;
  if (map_out) {
    op.r_out = RtRows{map_out->ptr, map_out->cstride, map_out->coff, fmt_};
  } else {
    y = new_tokens(rows, C, out_name);
    op.r_out = RtRows{y, C, 0, DT_F32};
  }
  op.rows = rows; op.C = C;
  op.w = upload(tensor(name + ".weight").data);
  op.bias = upload(tensor(name + ".bias").data);
  op.bytes = (double)rows * C * 8;
  ops_.push_back(op);
  return y;
}

namespace {
std::vector<float> rows_of(const HostTensor& t, int r0, int r1) {
  const size_t k = t.data.size() / (size_t)t.shape[0];
  return std::vector<float>(t.data.begin() + (size_t)r0 * k, t.data.begin() + (size_t)r1 * k);
}
std::vector<float> part_of(const std::vector<float>& v, int a, int b) { return std::vector<float>(v.begin() + a, v.begin() + b); }
}  // namespace

void RtDetr::build_graph() {
  const int S = lb_.net_h;
  img_ = new_view(S, S, 4);
  img_.plain = true;                                        // RGB0 bytes, really: [N][S][S][4] u8 in a buffer sized for fp32 (kept simple)
  if (fmt_ == DT_F32S) {
    sat_dev_ = (int*)alloc(sizeof(int));
    GTX_HIP(hipHostMalloc((void**)&h_sat_, sizeof(int)));
    *h_sat_ = 0;
  }
  if (has("rtdetr.meta")) {
    const auto& m = tensor("rtdetr.meta").data;
    GTX_CHECK(m.size() >= 4, "rtdetr.meta: [heads, points, queries, encoder heads] expected");
    nh_ = (int)m[0]; npts_ = (int)m[1]; nq_ = (int)m[2]; enc_heads_ = (int)m[3];
  }
  const std::string D = "model.28";
  nc_ = (int)tensor(D + ".enc_score_head.weight").shape[0];
  hd_ = (int)tensor(D + ".enc_score_head.weight").shape[1];
  ncp_ = (nc_ + 15) / 16 * 16;
  GTX_CHECK(nc_ == cfg_.nc, "RT-DETR score head has %d classes, the configuration says %d", nc_, cfg_.nc);
  GTX_CHECK(nq_ >= 1 && nq_ <= 512 && hd_ % nh_ == 0 && hd_ % 16 == 0, "RT-DETR decoder: %d queries, width %d, %d heads", nq_, hd_, nh_);
  ndl_ = 0;
  while (has(D + ".decoder.layers." + std::to_string(ndl_) + ".linear1.weight")) ++ndl_;
  GTX_CHECK(ndl_ >= 1, "RT-DETR: no decoder layers among the tensors");

  // ---- HGStem (model.0)
  const HostTensor& w1 = tensor("model.0.stem1.conv.weight");
  GTX_CHECK(w1.shape.size() == 4 && w1.shape[1] == 3 && w1.shape[2] == 3, "model.0.stem1 must be a 3x3 conv on 3 channels");
  const int cm0 = (int)w1.shape[0];
  GTX_CHECK(cm0 % 16 == 0, "model.0.stem1: %d output channels (a multiple of 16 is needed)", cm0);
  View s1 = new_view(S / 2, S / 2, cm0);
  {
    std::vector<float> w27((size_t)27 * cm0);
    for (int o = 0; o < cm0; ++o)
      for (int i = 0; i < 3; ++i)
        for (int t = 0; t < 9; ++t) w27[(size_t)(t * 3 + i) * cm0 + o] = w1.data[((size_t)o * 3 + i) * 9 + t];
    Op op;
    op.kind = Op::STEM1;
    op.name = "model.0.stem1.conv";
    op.family = "rt_stem1_kernel";
    op.a = img_.map(); op.b = s1.map();
    op.w = upload(w27);
    op.bias = upload(has("model.0.stem1.conv.bias") ? tensor("model.0.stem1.conv.bias").data : std::vector<float>(cm0, 0.f));
    op.flops = 2.0 * s1.h * s1.w * cm0 * 27;
    op.bytes = (double)S * S * 4 + (double)s1.h * s1.w * cm0 * 4;
    ops_.push_back(op);
    layer_views_[op.name] = s1;
  }
  View s2a = conv2x2("model.0.stem2a.conv", s1, nullptr);
  View cat_s = new_view(S / 2, S / 2, 2 * cm0);
  {
    Op op;
    op.kind = Op::POOL2;
    op.name = "model.0.pool";
    op.family = "rt_pool2_kernel";
    View dst = cat_s.slice(0, cm0);
    op.a = s1.map(); op.b = dst.map();
    op.bytes = (double)s1.h * s1.w * cm0 * 8;
    ops_.push_back(op);
  }
  {
    View dst = cat_s.slice(cm0, cm0);
    conv2x2("model.0.stem2b.conv", s2a, &dst);
  }
  View s3 = conv("model.0.stem3.conv", cat_s, 2, 2);
  int c1 = 0;
  View cat1 = hg_cat("model.1", S / 4, S / 4, &c1);
  {
    View dst = cat1.slice(0, c1);
    View s4 = conv("model.0.stem4.conv", s3, 1, 2, &dst);
    layer_views_["model.0"] = s4;
  }
  // ---- backbone
  View x1 = hgblock("model.1", cat1, c1, false, nullptr);
  int c3 = 0;
  View cat3 = hg_cat("model.3", S / 8, S / 8, &c3);
  dwconv("model.2.conv", x1, cat3.slice(0, c3), 2, 0);
  View x3 = hgblock("model.3", cat3, c3, false, nullptr);
  int c5 = 0, c6 = 0, c7 = 0, c9 = 0;
  View cat5 = hg_cat("model.5", S / 16, S / 16, &c5);
  dwconv("model.4.conv", x3, cat5.slice(0, c5), 2, 0);
  View cat6 = hg_cat("model.6", S / 16, S / 16, &c6);
  View cat7 = hg_cat("model.7", S / 16, S / 16, &c7);
  {
    View d6 = cat6.slice(0, c6), d7 = cat7.slice(0, c7);
    hgblock("model.5", cat5, c5, false, &d6);
    hgblock("model.6", cat6, c6, true, &d7);
  }
  View x7 = hgblock("model.7", cat7, c7, true, nullptr);
  View cat9 = hg_cat("model.9", S / 32, S / 32, &c9);
  dwconv("model.8.conv", x7, cat9.slice(0, c9), 2, 0);
  View x9 = hgblock("model.9", cat9, c9, false, nullptr);
  // ---- encoder: AIFI on P5 (model.10-11)
  View x10 = conv("model.10.conv", x9, 1, 0);
  const int E = x10.c, T5 = x10.h * x10.w;
  GTX_CHECK(E % enc_heads_ == 0 && E % 16 == 0, "AIFI: width %d with %d heads", E, enc_heads_);
  View cat27 = new_view(S / 32, S / 32, 2 * E);
  View x11 = new_view(x10.h, x10.w, E);
  {
    const std::string A = "model.11";
    // AIFI.build_2d_sincos_position_embedding(w, h, E): meshgrid(arange(w), arange(h), indexing="ij") flattened -- row t of the
    // embedding belongs to (i = t / h, j = t % h), whatever the token order of the map (upstream's own convention)
    std::vector<float> pos((size_t)T5 * E);
    const int pd = E / 4, pw = x10.w, ph = x10.h;
    for (int t = 0; t < T5; ++t) {
      const float gw = (float)(t / ph), gh = (float)(t % ph);
      for (int k = 0; k < pd; ++k) {
        const float omega = 1.f / std::pow(10000.f, (float)k / (float)pd);
        const float ow = gw * omega, oh = gh * omega;
        pos[(size_t)t * E + k] = std::sin(ow);
        pos[(size_t)t * E + pd + k] = std::cos(ow);
        pos[(size_t)t * E + 2 * pd + k] = std::sin(oh);
        pos[(size_t)t * E + 3 * pd + k] = std::cos(oh);
      }
    }
    (void)pw;
    Op op;
    op.kind = Op::TOKENS_IN;
    op.name = A + ".tokens";
    op.family = "rt_tokens_in_kernel";
    op.a = x10.map();
    op.p0 = upload(pos);
    float* src = new_tokens(T5, E, A + ".src");
    float* q = new_tokens(T5, E, A + ".q");
    op.p1 = src; op.p2 = q;
    op.bytes = (double)T5 * E * 12;
    ops_.push_back(op);
    const HostTensor &ipw = tensor(A + ".ma.in_proj_weight"), &ipb = tensor(A + ".ma.in_proj_bias");
    GTX_CHECK((int)ipw.shape[0] == 3 * E && (int)ipw.shape[1] == E, "AIFI in_proj shape");
    float* qkv = new_tokens(T5, 3 * E, A + ".qkv");
    linear(A + ".ma.in_proj.qk", rows_of(ipw, 0, 2 * E), part_of(ipb.data, 0, 2 * E), 2 * E, E, q, E, nullptr, T5, 0, nullptr, 0, qkv, 3 * E, "");
    linear(A + ".ma.in_proj.v", rows_of(ipw, 2 * E, 3 * E), part_of(ipb.data, 2 * E, 3 * E), E, E, src, E, nullptr, T5, 0, nullptr, 0, qkv + 2 * E, 3 * E, "");
    float* attn = new_tokens(T5, E, A + ".attn");
    {
      Op m;
      m.kind = Op::MHA;
      m.name = A + ".ma";
      m.family = E / enc_heads_ == 32 ? "rt_mha32_kernel" : "rt_mha_kernel";
      m.p0 = qkv; m.ld0 = 3 * E; m.p1 = attn; m.ld1 = E;
      m.T = T5; m.C = E; m.heads = enc_heads_;
      m.flops = 4.0 * T5 * (double)T5 * E;
      m.bytes = (double)T5 * E * 16;
      ops_.push_back(m);
    }
    float* t1 = linear(A + ".ma.out_proj", tensor(A + ".ma.out_proj.weight").data, tensor(A + ".ma.out_proj.bias").data, E, E, attn, E, nullptr, T5, 0, src, E, nullptr, 0, A + ".t1");
    float* n1 = layernorm_tokens(A + ".norm1", t1, T5, E, A + ".norm1");
    const int dff = (int)tensor(A + ".fc1.weight").shape[0];
    float* ff = linear(A + ".fc1", tensor(A + ".fc1.weight").data, tensor(A + ".fc1.bias").data, dff, E, n1, E, nullptr, T5, 3, nullptr, 0, nullptr, 0, A + ".ff");
    float* t2 = linear(A + ".fc2", tensor(A + ".fc2.weight").data, tensor(A + ".fc2.bias").data, E, dff, ff, (dff + 15) / 16 * 16, nullptr, T5, 0, n1, E, nullptr, 0, A + ".t2");
    layernorm_tokens(A + ".norm2", t2, T5, E, "", &x11);
    layer_views_[A] = x11;
  }
  // ---- encoder: CCFM (model.12-27)
  {
    View d = cat27.slice(E, E);
    conv("model.12.conv", x11, 1, 1, &d);
  }
  const View x12 = cat27.slice(E, E);
  View cat16 = new_view(S / 16, S / 16, 2 * E);
  upsample("model.13", x12, cat16.slice(0, E));
  {
    View d = cat16.slice(E, E);
    conv("model.14.conv", x7, 1, 0, &d);
  }
  View x16 = repc3("model.16", cat16);
  View cat24 = new_view(S / 16, S / 16, 2 * E);
  {
    View d = cat24.slice(E, E);
    conv("model.17.conv", x16, 1, 1, &d);
  }
  const View x17 = cat24.slice(E, E);
  View cat21 = new_view(S / 8, S / 8, 2 * E);
  upsample("model.18", x17, cat21.slice(0, E));
  {
    View d = cat21.slice(E, E);
    conv("model.19.conv", x3, 1, 0, &d);
  }
  View x21 = repc3("model.21", cat21);
  {
    View d = cat24.slice(0, E);
    conv("model.22.conv", x21, 2, 1, &d);
  }
  View x24 = repc3("model.24", cat24);
  {
    View d = cat27.slice(0, E);
    conv("model.25.conv", x24, 2, 1, &d);
  }
  View x27 = repc3("model.27", cat27);

  // ---- RTDETRDecoder (model.28)
  const View feat[3] = {x21, x24, x27};
  const int LP = 3 * npts_;
  View proj[3], val[3], enc[3], score[3];
  std::vector<float> wv((size_t)ndl_ * hd_ * hd_), bv((size_t)ndl_ * hd_);     // the six layers' value_proj stacked: one conv per level
  for (int i = 0; i < ndl_; ++i) {
    const std::string lp = D + ".decoder.layers." + std::to_string(i) + ".cross_attn.value_proj";
    const HostTensor &w = tensor(lp + ".weight"), &b = tensor(lp + ".bias");
    GTX_CHECK((int)w.shape[0] == hd_ && (int)w.shape[1] == hd_, "%s shape", lp.c_str());
    memcpy(&wv[(size_t)i * hd_ * hd_], w.data.data(), sizeof(float) * hd_ * hd_);
    memcpy(&bv[(size_t)i * hd_], b.data.data(), sizeof(float) * hd_);
  }
  for (int l = 0; l < 3; ++l) {
    const std::string ip = D + ".input_proj." + std::to_string(l) + ".0";
    proj[l] = conv(ip, feat[l], 1, 0);
    GTX_CHECK(proj[l].c == hd_, "%s: %d output channels, the decoder is %d wide", ip.c_str(), proj[l].c, hd_);
    layer_views_[D + ".feats." + std::to_string(l)] = proj[l];
  }
  for (int l = 0; l < 3; ++l) val[l] = conv_raw(D + ".value_proj." + std::to_string(l), wv, ndl_ * hd_, hd_, 1, bv.data(), proj[l], 1, 0, nullptr, nullptr);
  for (int l = 0; l < 3; ++l) {                            // `valid_mask * feats`: after the values, which read the unmasked rows
    Op op;
    op.kind = Op::MASK;
    op.name = D + ".valid_mask." + std::to_string(l);
    op.family = "rt_mask_invalid_kernel";
    op.a = proj[l].map();
    op.level = l;
    ops_.push_back(op);
  }
  {
    const HostTensor &w = tensor(D + ".enc_output.0.weight"), &b = tensor(D + ".enc_output.0.bias");
    for (int l = 0; l < 3; ++l) enc[l] = conv_raw(D + ".enc_output.0." + std::to_string(l), w.data, hd_, hd_, 1, b.data.data(), proj[l], 1, 0, nullptr, nullptr);
    const float* g = upload(tensor(D + ".enc_output.1.weight").data);
    const float* be = upload(tensor(D + ".enc_output.1.bias").data);
    for (int l = 0; l < 3; ++l) {
      Op op;
      op.kind = Op::LAYERNORM;
      op.name = D + ".enc_output.1." + std::to_string(l);
      op.family = "rt_layernorm_kernel";
      op.r_in = op.r_out = RtRows{enc[l].ptr, enc[l].cstride, enc[l].coff, fmt_};
      op.rows = (long)enc[l].h * enc[l].w; op.C = hd_;
      op.w = g; op.bias = be;
      op.bytes = (double)op.rows * hd_ * 8;
      ops_.push_back(op);
      layer_views_[D + ".enc_output." + std::to_string(l)] = enc[l];
    }
    const HostTensor &ws = tensor(D + ".enc_score_head.weight"), &bs = tensor(D + ".enc_score_head.bias");
    std::vector<float> wp((size_t)ncp_ * hd_, 0.f), bp(ncp_, 0.f);
    memcpy(wp.data(), ws.data.data(), sizeof(float) * nc_ * hd_);
    memcpy(bp.data(), bs.data.data(), sizeof(float) * nc_);
    for (int l = 0; l < 3; ++l) {
      score[l] = conv_raw(D + ".enc_score_head." + std::to_string(l), wp, ncp_, hd_, 1, bp.data(), enc[l], 1, 0, nullptr, nullptr, true);
      layer_views_[D + ".enc_scores." + std::to_string(l)] = score[l];
    }
  }
  int S_total = 0;
  for (int l = 0; l < 3; ++l) S_total += proj[l].h * proj[l].w;
  GTX_CHECK(S_total >= nq_, "RT-DETR: %d anchors for %d queries", S_total, nq_);
  int* topk_idx = (int*)alloc(sizeof(int) * cfg_.max_batch * nq_);
  {
    Op op;
    op.kind = Op::TOPK;
    op.name = D + ".topk";
    op.family = "rt_topk_kernel";
    for (int l = 0; l < 3; ++l) { op.lv.ptr[l] = score[l].ptr; op.lv.h[l] = score[l].h; op.lv.w[l] = score[l].w; op.lv.cstride[l] = score[l].cstride; op.lv.coff[l] = score[l].coff; }
    op.lv.n_levels = 3;
    op.p1 = (float*)alloc(sizeof(unsigned) * cfg_.max_batch * (size_t)S_total);   // key scratch
    op.p2 = (float*)topk_idx;
    op.bytes = (double)S_total * ncp_ * 4;
    ops_.push_back(op);
  }
  float* embed = new_tokens(nq_, hd_, D + ".embed");
  float* anchors = new_tokens(nq_, 4, D + ".anchors");
  {
    Op op;
    op.kind = Op::GATHER;
    op.name = D + ".gather";
    op.family = "rt_gather_kernel";
    for (int l = 0; l < 3; ++l) { op.lv.ptr[l] = enc[l].ptr; op.lv.h[l] = enc[l].h; op.lv.w[l] = enc[l].w; op.lv.cstride[l] = enc[l].cstride; op.lv.coff[l] = enc[l].coff; }
    op.lv.n_levels = 3;
    op.p0 = (const float*)topk_idx; op.p1 = embed; op.p2 = anchors;
    ops_.push_back(op);
  }
  layer_views_[D + ".topk"] = View{topk_idx, cfg_.max_batch, 1, nq_, 1, 0, 1, true};
  refer_ = new_tokens(nq_, 16, D + ".refer");
  auto lin_named = [&](const std::string& name, const float* x, int ldx, const float* x2, int act, const float* res, int ldr, const std::string& out_name) {
    const HostTensor& w = tensor(name + ".weight");
    return linear(name, w.data, tensor(name + ".bias").data, (int)w.shape[0], (int)w.shape[1], x, ldx, x2, nq_, act, res, ldr, nullptr, 0, out_name);
  };
  auto bbox_head = [&](const std::string& name, const float* x, int mode) {
    float* h1 = lin_named(name + ".layers.0", x, hd_, nullptr, 2, nullptr, 0, "");
    float* h2 = lin_named(name + ".layers.1", h1, hd_, nullptr, 2, nullptr, 0, "");
    float* dl = lin_named(name + ".layers.2", h2, hd_, nullptr, 0, nullptr, 0, name + ".delta");   // 4 outputs in 16 columns
    Op op;
    op.kind = Op::REFER;
    op.name = name + ".refer";
    op.family = "rt_refer_kernel";
    op.p0 = dl; op.ld0 = 16; op.p1 = anchors; op.p2 = refer_; op.mode = mode;
    ops_.push_back(op);
  };
  bbox_head(D + ".enc_bbox_head", embed, 0);
  const float* out = embed;
  RtLevels vlv{};
  for (int l = 0; l < 3; ++l) { vlv.ptr[l] = val[l].ptr; vlv.h[l] = val[l].h; vlv.w[l] = val[l].w; vlv.cstride[l] = val[l].cstride; }
  vlv.n_levels = 3;
  for (int i = 0; i < ndl_; ++i) {
    const std::string lp = D + ".decoder.layers." + std::to_string(i);
    float* qp1 = lin_named(D + ".query_pos_head.layers.0", refer_, 16, nullptr, 2, nullptr, 0, "");
    float* qpos = lin_named(D + ".query_pos_head.layers.1", qp1, 2 * hd_, nullptr, 0, nullptr, 0, "");
    // self attention over the queries
    const HostTensor &ipw = tensor(lp + ".self_attn.in_proj_weight"), &ipb = tensor(lp + ".self_attn.in_proj_bias");
    GTX_CHECK((int)ipw.shape[0] == 3 * hd_ && (int)ipw.shape[1] == hd_, "%s.self_attn.in_proj shape", lp.c_str());
    float* qkv = new_tokens(nq_, 3 * hd_, "");
    if ((2 * hd_) % 64 == 0) {                       // q, k from query + query_pos and v from the query alone, one launch
      linear(lp + ".self_attn.in_proj", ipw.data, ipb.data, 3 * hd_, hd_, out, hd_, qpos, nq_, 0, nullptr, 0, qkv, 3 * hd_, "");
      ops_.back().lin.x2_cols = 2 * hd_;
    } else {
      linear(lp + ".self_attn.in_proj.qk", rows_of(ipw, 0, 2 * hd_), part_of(ipb.data, 0, 2 * hd_), 2 * hd_, hd_, out, hd_, qpos, nq_, 0, nullptr, 0, qkv, 3 * hd_, "");
      linear(lp + ".self_attn.in_proj.v", rows_of(ipw, 2 * hd_, 3 * hd_), part_of(ipb.data, 2 * hd_, 3 * hd_), hd_, hd_, out, hd_, nullptr, nq_, 0, nullptr, 0, qkv + 2 * hd_, 3 * hd_, "");
    }
    float* attn = new_tokens(nq_, hd_, "");
    {
      Op m;
      m.kind = Op::MHA;
      m.name = lp + ".self_attn";
      m.family = hd_ / nh_ == 32 ? "rt_mha32_kernel" : "rt_mha_kernel";
      m.p0 = qkv; m.ld0 = 3 * hd_; m.p1 = attn; m.ld1 = hd_;
      m.T = nq_; m.C = hd_; m.heads = nh_;
      m.flops = 4.0 * nq_ * (double)nq_ * hd_;
      ops_.push_back(m);
    }
    float* t1 = lin_named(lp + ".self_attn.out_proj", attn, hd_, nullptr, 0, out, hd_, "");
    float* o1 = layernorm_tokens(lp + ".norm1", t1, nq_, hd_, lp + ".norm1");
    // cross attention: sampling offsets and attention weights in one linear layer, then the deformable sampling
    std::vector<float> wo = tensor(lp + ".cross_attn.sampling_offsets.weight").data, bo = tensor(lp + ".cross_attn.sampling_offsets.bias").data;
    const auto &wa = tensor(lp + ".cross_attn.attention_weights.weight").data, &ba = tensor(lp + ".cross_attn.attention_weights.bias").data;
    GTX_CHECK((int)bo.size() == nh_ * LP * 2 && (int)ba.size() == nh_ * LP, "%s.cross_attn: %d heads x 3 levels x %d points expected", lp.c_str(), nh_, npts_);
    wo.insert(wo.end(), wa.begin(), wa.end());
    bo.insert(bo.end(), ba.begin(), ba.end());
    GTX_CHECK((nh_ * LP * 3) % 16 == 0, "cross attention: %d outputs (a multiple of 16 is needed)", nh_ * LP * 3);
    float* offaw = linear(lp + ".cross_attn.offsets+weights", wo, bo, nh_ * LP * 3, hd_, o1, hd_, qpos, nq_, 0, nullptr, 0, nullptr, 0, "");
    float* samp = new_tokens(nq_, hd_, lp + ".cross_attn.sampled");
    {
      Op d;
      d.kind = Op::DEFORM;
      d.name = lp + ".cross_attn";
      d.family = "rt_deform_kernel";
      d.lv = vlv;
      for (int l = 0; l < 3; ++l) d.lv.coff[l] = val[l].coff + i * hd_;
      d.p0 = offaw; d.p1 = samp; d.p2 = refer_;
      d.bytes = (double)nq_ * hd_ * LP * 4 * 4;
      ops_.push_back(d);
    }
    float* t2 = lin_named(lp + ".cross_attn.output_proj", samp, hd_, nullptr, 0, o1, hd_, "");
    float* o2 = layernorm_tokens(lp + ".norm2", t2, nq_, hd_, lp + ".norm2");
    const int dff = (int)tensor(lp + ".linear1.weight").shape[0];
    float* f1 = lin_named(lp + ".linear1", o2, hd_, nullptr, 2, nullptr, 0, "");
    float* t3 = lin_named(lp + ".linear2", f1, (dff + 15) / 16 * 16, nullptr, 0, o2, hd_, "");
    float* o3 = layernorm_tokens(lp + ".norm3", t3, nq_, hd_, lp);
    bbox_head(D + ".dec_bbox_head." + std::to_string(i), o3, 1);
    out = o3;
  }
  {
    const std::string sh = D + ".dec_score_head." + std::to_string(ndl_ - 1);
    const HostTensor& w = tensor(sh + ".weight");
    GTX_CHECK((int)w.shape[0] == nc_, "%s: %d classes", sh.c_str(), (int)w.shape[0]);
    logits_ = linear(sh, w.data, tensor(sh + ".bias").data, nc_, hd_, out, hd_, nullptr, nq_, 0, nullptr, 0, nullptr, 0, D + ".logits");
  }
}

void RtDetr::set_batch(int nb) {
  if (nb == cur_nb_) return;
  for (Op& op : ops_) {
    if (op.kind != Op::CONV) continue;
    for (int i = 0; i < op.grp.count; ++i) op.grp.p[i].N = nb;
    conv_group_finalize(op.grp, op.cfg);
    const ConvProblem& p = op.grp.p[0];
    op.flops = conv_flops(p, op.cfg.ks) / nb;
    op.bytes = ((double)p.H * p.W * p.Cin + (double)p.Ho * p.Wo * p.Cout) * 4 + (double)p.Cout * p.Cin * op.cfg.ks * op.cfg.ks * 4 / nb;
  }
  cur_nb_ = nb;
}

void RtDetr::finalize() {
  GTX_CHECK(!finalized_, "finalize called twice");
  GTX_HIP(hipSetDevice(ctx_->device));
  build_graph();
  const int N = cfg_.max_batch;
  gray_h_ = cfg_.frame_h / 2;
  gray_w_ = cfg_.frame_w / 2;
  gray_.alloc((size_t)kGrayRing * N * gray_h_ * gray_w_);
  class_mask_[0] = class_mask_[1] = cfg_.n_classes == 0 ? ~0ull : 0ull;
  for (int i = 0; i < cfg_.n_classes; ++i)
    if (cfg_.classes[i] >= 0 && cfg_.classes[i] < 128) class_mask_[cfg_.classes[i] >> 6] |= 1ull << (cfg_.classes[i] & 63);
  raw_ = (float*)alloc(sizeof(float) * N * nq_ * (4 + nc_));
  out_rows_ = (float*)alloc(sizeof(float) * 6 * N * cfg_.max_det);
  out_n_ = (int*)alloc(sizeof(int) * N);
  GTX_HIP(hipHostMalloc((void**)&h_out_n_, sizeof(int) * N));
  GTX_HIP(hipHostMalloc((void**)&h_out_rows_, sizeof(float) * 6 * N * cfg_.max_det));
  if (fmt_ != DT_F32S || !env_on("GTX_SAT_FALLBACK", true)) tensors_.clear();   // the split path keeps the host copies for fall_back_to_exact
  set_batch(1);
  GTX_HIP(hipStreamSynchronize(ctx_->stream));
  finalized_ = true;
}

void RtDetr::run_op(const Op& op, int nb, hipStream_t s) {
  switch (op.kind) {
    case Op::CONV: conv_launch(op.grp, op.cfg, s); break;
    case Op::STEM1: launch_rt_stem1(fmt_, op.a.ptr, nb, op.a.h, op.a.w, op.w, op.bias, op.b, sat_dev_, s); break;
    case Op::POOL2: launch_rt_pool2(fmt_, op.a, op.b, nb, sat_dev_, s); break;
    case Op::DWCONV: launch_rt_dwconv(fmt_, op.a, op.b, nb, op.k, op.stride, op.w, op.bias, op.act, sat_dev_, s); break;
    case Op::UPSAMPLE: launch_rt_upsample2x(fmt_, op.a, op.b, nb, s); break;
    case Op::TOKENS_IN: launch_rt_tokens_in(fmt_, op.a, nb, op.p0, op.p1, op.p2, s); break;
    case Op::LINEAR: {
      RtLinear L = op.lin;
      L.M = op.lin.M * nb;
      launch_rt_linear(L, s);
      break;
    }
    case Op::LAYERNORM: launch_rt_layernorm(op.r_in, op.r_out, op.rows * nb, op.C, op.w, op.bias, sat_dev_, s); break;
    case Op::MHA: launch_rt_mha(op.p0, op.ld0, nb, op.T, op.C, op.heads, op.p1, op.ld1, s); break;
    case Op::MASK: launch_rt_mask_invalid(fmt_, op.a, nb, op.level, s); break;
    case Op::TOPK: launch_rt_topk(fmt_ == DT_F16 ? DT_F16 : DT_F32, op.lv, nc_, nb, nq_, (unsigned*)op.p1, (int*)op.p2, s); break;
    case Op::GATHER: launch_rt_gather(fmt_, op.lv, hd_, nb, nq_, (const int*)op.p0, op.p1, op.p2, s); break;
    case Op::REFER: launch_rt_refer(op.p0, op.ld0, op.p1, op.p2, nb * nq_, op.mode, s); break;
    case Op::DEFORM: launch_rt_deform(fmt_, op.lv, hd_, nh_, npts_, op.p0, op.p2, nb, nq_, op.p1, s); break;
  }
}

void RtDetr::run_forward(int nb, hipStream_t s, bool traced) {
  if (!traced) {
    for (const Op& op : ops_) run_op(op, nb, s);
    return;
  }
  for (size_t i = 0; i < ops_.size(); ++i) {
    GTX_HIP(hipEventRecord(trace_ev_[i], s));
    run_op(ops_[i], nb, s);
  }
  GTX_HIP(hipEventRecord(trace_ev_[ops_.size()], s));
}

void RtDetr::set_trace(int every_n) {
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

void RtDetr::trace_report(std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms, std::vector<double>& flops,
                          std::vector<double>& bytes) {
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

void RtDetr::fall_back_to_exact() {
  gtx_det_config c = cfg_;
  c.fp32_split = 0;
  std::unique_ptr<RtDetr> d(new RtDetr(ctx_, c));
  for (const auto& kv : tensors_) d->set_tensor(kv.first, kv.second.data.data(), (int)kv.second.shape.size(), kv.second.shape.data());
  d->finalize();
  if (trace_every_ > 0) d->set_trace(trace_every_);
  GTX_HIP(hipStreamSynchronize(ctx_->stream));
  ops_.clear();
  layer_views_.clear();
  bufs_.clear();
  tensors_.clear();
  exact_ = std::move(d);
}

bool RtDetr::saturated(bool clear) {
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

void RtDetr::submit_dev(const void* frames, int nb, int h, int w) {
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
  launch_preprocess(DT_F32, (const uint8_t*)frames, nb, lb_, img_.ptr, gray, gray_h_, gray_w_, s);
  GTX_HIP(hipEventRecord(ev_[1], s));
  flight_traced_ = trace_every_ > 0 && (trace_count_++ % trace_every_) == 0;
  run_forward(nb, s, flight_traced_);
  GTX_HIP(hipEventRecord(ev_[2], s));
  launch_rt_post(logits_, ncp_, refer_, nb, nq_, nc_, cfg_.conf, class_mask_, cfg_.frame_w, cfg_.frame_h, cfg_.max_det, out_rows_, out_n_, raw_, s);
  GTX_HIP(hipMemcpyAsync(h_out_n_, out_n_, sizeof(int) * nb, hipMemcpyDeviceToHost, s));
  GTX_HIP(hipMemcpyAsync(h_out_rows_, out_rows_, sizeof(float) * 6 * nb * cfg_.max_det, hipMemcpyDeviceToHost, s));
  if (sat_dev_) GTX_HIP(hipMemcpyAsync(h_sat_, sat_dev_, sizeof(int), hipMemcpyDeviceToHost, s));
  GTX_HIP(hipEventRecord(ev_[3], s));
  in_flight_ = true;
  flight_nb_ = nb;
}

void RtDetr::collect(int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) {
  if (exact_) return exact_->collect(n_out, xyxy, conf, cls, speed_ms);
  GTX_CHECK(in_flight_, "collect without a submitted batch");
  GTX_HIP(hipSetDevice(ctx_->device));
  GTX_HIP(hipEventSynchronize(ev_[3]));
  in_flight_ = false;
  collected_gray_slot_ = gray_slot_;
  if (h_sat_ && *h_sat_) {
    sat_seen_ = true;
    if (fmt_ == DT_F32S && !tensors_.empty()) {        // this batch again at fp32's range, and every later one (Detector::collect's rule)
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
      trace_flops_[i] += ops_[i].flops * flight_nb_;
      trace_bytes_[i] += ops_[i].bytes * flight_nb_;
    }
    flight_traced_ = false;
  }
  for (int b = 0; b < flight_nb_; ++b) {
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
  if (speed_ms)
    for (int i = 0; i < 3; ++i) GTX_HIP(hipEventElapsedTime(&speed_ms[i], ev_[i], ev_[i + 1]));
}

void RtDetr::detect_dev(const void* frames, int nb, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) {
  submit_dev(frames, nb, h, w);
  collect(n_out, xyxy, conf, cls, speed_ms);
}

void RtDetr::detect_host(const uint8_t* frame, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) {
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
    speed_ms[0] += up_ms;
  }
}

const void* RtDetr::gray(int b, int* gh, int* gw) const {
  if (exact_) return exact_->gray(b, gh, gw);
  if (gh) *gh = gray_h_;
  if (gw) *gw = gray_w_;
  if (b < 0 || b >= cfg_.max_batch) return nullptr;
  return gray_.as<uint8_t>() + ((size_t)collected_gray_slot_ * cfg_.max_batch + b) * gray_h_ * gray_w_;
}

void RtDetr::raw_output(int b, float* out, int* n_anchors, bool logits) {
  if (exact_) return exact_->raw_output(b, out, n_anchors, logits);
  GTX_CHECK(finalized_ && cur_nb_ > 0 && b >= 0 && b < cur_nb_, "raw_output: no forward pass for slot %d", b);
  GTX_CHECK(!in_flight_, "raw_output while a batch is in flight: call collect first");
  const size_t per = (size_t)nq_ * (4 + nc_);
  GTX_HIP(hipMemcpy(out, raw_ + per * b, per * sizeof(float), hipMemcpyDeviceToHost));
  if (logits) {
    std::vector<float> lg((size_t)nq_ * ncp_);
    GTX_HIP(hipMemcpy(lg.data(), logits_ + (size_t)b * nq_ * ncp_, lg.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int q = 0; q < nq_; ++q)
      for (int c = 0; c < nc_; ++c) out[(size_t)q * (4 + nc_) + 4 + c] = lg[(size_t)q * ncp_ + c];
  }
  if (n_anchors) *n_anchors = nq_;
}

void RtDetr::layer_output(int b, const std::string& layer, float* out, int* h, int* w, int* c) {
  if (exact_) return exact_->layer_output(b, layer, out, h, w, c);
  GTX_CHECK(!(out && in_flight_), "layer_output while a batch is in flight: call collect first");
  auto it = layer_views_.find(layer);
  if (it == layer_views_.end()) fail(-1, "unknown layer '%s'", layer.c_str());
  const View& v = it->second;
  if (h) *h = v.h;
  if (w) *w = v.w;
  if (c) *c = v.c;
  if (!out) return;
  GTX_CHECK(b >= 0 && b < cfg_.max_batch, "bad batch slot");
  const size_t px = (size_t)v.h * v.w;
  const size_t es = (fmt_ == DT_F16 && !v.plain) ? 2 : 4;
  std::vector<uint8_t> host(px * v.cstride * es);
  GTX_HIP(hipMemcpy(host.data(), (const uint8_t*)v.ptr + (size_t)b * px * v.cstride * es, host.size(), hipMemcpyDeviceToHost));
  for (size_t p = 0; p < px; ++p)
    for (int k = 0; k < v.c; ++k) {
      const size_t src = p * v.cstride + v.coff + k;
      float f;
      if (fmt_ == DT_F32S && !v.plain) f = pair_element(host.data(), src);
      else if (es == 2) { _Float16 hv; memcpy(&hv, host.data() + src * 2, 2); f = (float)hv; }
      else memcpy(&f, host.data() + src * 4, 4);
      out[p * v.c + k] = f;
    }
}

void RtDetr::profile(int nb, int iters, std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms,
                     std::vector<double>& flops, std::vector<double>& bytes) {
  if (exact_) return exact_->profile(nb, iters, names, launches, ms, flops, bytes);
  GTX_CHECK(finalized_, "detector not finalized");
  GTX_CHECK(nb >= 1 && nb <= cfg_.max_batch && iters >= 1, "bad profile arguments");
  hipStream_t s = ctx_->stream;
  set_batch(nb);
  std::vector<hipEvent_t> ev(ops_.size() + 1);
  for (auto& e : ev) GTX_HIP(hipEventCreate(&e));
  std::map<std::string, size_t> idx;
  const bool per_op = std::getenv("GTX_PROFILE_PER_OP") != nullptr;
  auto slot = [&](const std::string& fam) {
    auto it = idx.find(fam);
    if (it != idx.end()) return it->second;
    idx[fam] = names.size();
    names.push_back(fam);
    launches.push_back(0); ms.push_back(0.f); flops.push_back(0.0); bytes.push_back(0.0);
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
      const size_t k = slot(per_op ? (i < 10 ? "00" : i < 100 ? "0" : "") + std::to_string(i) + " " + ops_[i].name : ops_[i].family);
      launches[k] += 1;
      ms[k] += t;
      flops[k] += ops_[i].flops * nb;
      bytes[k] += ops_[i].bytes * nb;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
}

}  // namespace gtx
