// RT-DETR detector runtime (rtdetr-l topology; widths, class count and layer counts read off the tensors): builds the layer
// graph from ultralytics-named fused tensors and runs preprocess -> HGNetv2 backbone -> AIFI + CCFM encoder -> query
// selection -> deformable-attention decoder -> score / box stage on a HIP stream. Stands in for what ultralytics' RTDETR
// predictor does underneath model.track() when the model's yaml names RT-DETR (geotrax/extract.py:222-225, :153).
//
// Convolutions (and the per-anchor linear layers, which are 1x1 convolutions on the three feature levels) run on the
// detector's MFMA kernels in the activation format of the run (split-f16x3 pair format by default, exact fp32 with
// fp32_split = 0); everything on AIFI's tokens and the decoder's 300 queries runs at exact fp32 (rtdetr_kernels.hpp).
#pragma once
#include "detector.hpp"
#include "rtdetr_kernels.hpp"

namespace gtx {

class RtDetr : public DetectorBase {
 public:
  RtDetr(gtx_ctx* ctx, const gtx_det_config& cfg);
  ~RtDetr() override;
  void set_tensor(const std::string& name, const float* data, int ndim, const int64_t* shape) override;
  void finalize() override;
  void input_size(int* h, int* w) const override { *h = lb_.net_h; *w = lb_.net_w; }
  void detect_dev(const void* frames, int nb, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) override;
  void submit_dev(const void* frames, int nb, int h, int w) override;
  void collect(int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) override;
  void detect_host(const uint8_t* frame, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) override;
  const void* gray(int b, int* gh, int* gw) const override;
  // [nq][4 + nc]: xywh normalised to the frame + class scores of every query of image b (logits: the pre-sigmoid class logits)
  void raw_output(int b, float* out, int* n_anchors, bool logits = false) override;
  void layer_output(int b, const std::string& layer, float* out, int* h, int* w, int* c) override;
  void profile(int nb, int iters, std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms,
               std::vector<double>& flops, std::vector<double>& bytes) override;
  void set_trace(int every_n) override;
  void trace_report(std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms, std::vector<double>& flops,
                    std::vector<double>& bytes) override;
  void features(int, float*, int, int*, int*) const override { fail(-3, "RT-DETR: appearance vectors (obj_feats) are not implemented"); }
  bool saturated(bool clear) override;
  bool fell_back() const override { return exact_ != nullptr; }
  void pad_skip(int* on, int* skipped, int* total) const override { if (on) *on = 0; if (skipped) *skipped = 0; if (total) *total = 0; }
  void sparse_box(int* on, int* overflows) const override { if (on) *on = 0; if (overflows) *overflows = 0; }

 private:
  struct Op {
    enum Kind { CONV, STEM1, POOL2, DWCONV, UPSAMPLE, TOKENS_IN, LINEAR, LAYERNORM, MHA, MASK, TOPK, GATHER, REFER, DEFORM } kind = CONV;
    std::string name, family;
    ConvGroup grp{};
    ConvConfig cfg{};
    RtMap a{}, b{};                  // map in / out
    const float* w = nullptr;        // STEM1 / DWCONV weights
    const float* bias = nullptr;
    int k = 0, stride = 1, act = 0, level = 0, mode = 0;
    RtLinear lin{};                  // LINEAR (M = rows per image; scaled by the batch at launch)
    RtRows r_in{}, r_out{};          // LAYERNORM
    long rows = 0;                   // LAYERNORM: rows per image
    int C = 0, heads = 0, T = 0;     // LAYERNORM width / MHA
    const float* p0 = nullptr;       // TOKENS_IN pos; MHA qkv; REFER delta; DEFORM offaw; GATHER idx (int)
    float* p1 = nullptr;             // TOKENS_IN src; MHA out; REFER anchors; DEFORM out; GATHER embed
    float* p2 = nullptr;             // TOKENS_IN q; REFER refer; DEFORM refer; GATHER anchors
    int ld0 = 0, ld1 = 0;
    RtLevels lv{};                   // TOPK scores / GATHER enc / DEFORM values
    double flops = 0, bytes = 0;     // per image
  };
  struct View {
    void* ptr = nullptr;
    int n = 0, h = 0, w = 0, cstride = 0, coff = 0, c = 0;
    bool plain = false;              // plain fp32 whatever the activation format (token rows, score maps)
    View slice(int off, int cnt) const { View v = *this; v.coff = coff + off; v.c = cnt; return v; }
    RtMap map() const { return RtMap{ptr, h, w, cstride, coff, c}; }
  };

  void* alloc(size_t bytes);
  float* upload(const std::vector<float>& v);
  View new_view(int h, int w, int c, bool plain = false);
  float* new_tokens(int rows_per_image, int ld, const std::string& name);
  const HostTensor& tensor(const std::string& name) const;
  bool has(const std::string& name) const { return tensors_.count(name) != 0; }
  // graph building
  View conv_raw(const std::string& name, const std::vector<float>& w_oihw, int cout, int cin, int ks, const float* bias_host, const View& x, int stride,
                int act, const View* out_slice, const View* residual, bool plain_out = false);
  View conv(const std::string& name, const View& x, int stride, int act, const View* out_slice = nullptr, const View* residual = nullptr);
  View conv2x2(const std::string& name, const View& x, const View* out_slice);
  void dwconv(const std::string& name, const View& x, const View& out, int stride, int act);
  View hg_cat(const std::string& pfx, int h, int w, int* c1);
  View hgblock(const std::string& pfx, const View& cat, int c1, bool shortcut, const View* out_slice);
  View repc3(const std::string& pfx, const View& x);
  void upsample(const std::string& name, const View& src, const View& dst);
  float* linear(const std::string& name, const std::vector<float>& w, const std::vector<float>& bias, int nout, int k, const float* x, int ldx, const float* x2,
                int rows, int act, const float* res, int ldr, float* y, int ldy, const std::string& out_name);
  float* layernorm_tokens(const std::string& name, const float* x, int rows, int C, const std::string& out_name, const View* map_out = nullptr);
  void build_graph();
  void run_op(const Op& op, int nb, hipStream_t s);
  void run_forward(int nb, hipStream_t s, bool traced);
  void set_batch(int nb);
  void fall_back_to_exact();

  gtx_ctx* ctx_;
  gtx_det_config cfg_;
  std::unique_ptr<RtDetr> exact_;
  int fmt_;                          // activation format of the map tensors: DT_F32 or DT_F32S
  Letterbox lb_{};
  std::map<std::string, HostTensor> tensors_;
  std::vector<DevBuf> bufs_;
  std::vector<Op> ops_;
  std::map<std::string, View> layer_views_;
  bool finalized_ = false;
  int cur_nb_ = 0;
  int nh_ = 8, npts_ = 4, nq_ = 300, enc_heads_ = 8, hd_ = 256, nc_ = 0, ncp_ = 0, ndl_ = 0;

  View img_;
  DevBuf frame_stage_, gray_;
  int gray_h_ = 0, gray_w_ = 0;
  const void* cur_frames_ = nullptr;
  static constexpr int kGrayRing = 16;
  int gray_slot_ = 0, collected_gray_slot_ = 0;
  bool in_flight_ = false;
  int flight_nb_ = 0;
  int* sat_dev_ = nullptr;
  int* h_sat_ = nullptr;
  bool sat_seen_ = false;
  // post stage
  const float* logits_ = nullptr;    // [N * nq][ncp]
  float* refer_ = nullptr;           // [N * nq][16]
  float* raw_ = nullptr;             // [N][nq][4 + nc]
  unsigned long long class_mask_[2] = {~0ull, ~0ull};
  float* out_rows_ = nullptr;
  int* out_n_ = nullptr;
  int* h_out_n_ = nullptr;
  float* h_out_rows_ = nullptr;
  hipEvent_t ev_[4]{};
  hipEvent_t ev_up_[2]{};
  int trace_every_ = 0, trace_count_ = 0;
  bool flight_traced_ = false;
  std::vector<hipEvent_t> trace_ev_;
  std::vector<double> trace_ms_, trace_flops_, trace_bytes_;
  std::vector<int> trace_n_;
};

}  // namespace gtx
