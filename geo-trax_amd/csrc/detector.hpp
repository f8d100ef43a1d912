// YOLOv8 detector runtime: builds the layer graph from ultralytics-named fused tensors, plans
// the NHWC buffers (concats are channel slices, never copies), and runs
// preprocess -> forward -> decode -> NMS on a HIP stream. Stands in for what
// ultralytics' predictor does underneath model.track() (geotrax/extract.py:153).
#pragma once
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/gtx.h"
#include "common.hpp"
#include "conv_igemm.hpp"
#include "det_kernels.hpp"

struct gtx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop{};
  ~gtx_ctx() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

namespace gtx {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// A channel slice of an NHWC device buffer.
struct View {
  void* ptr = nullptr;
  int n = 0, h = 0, w = 0;
  int cstride = 0, coff = 0, c = 0;
  bool plain = false;      // split-f16x3 path: plain fp32 instead of the pair format (the Detect head's last stage)
  View slice(int off, int cnt) const {
    View v = *this;
    v.coff = coff + off;
    v.c = cnt;
    return v;
  }
};

struct Op {
  enum Kind { CONV, STEM, POOL, UPSAMPLE } kind = CONV;
  std::string name;      // ultralytics module path ("model.2.m.0.cv1") or group label
  std::string family;    // kernel symbol, as rocprof prints it
  ConvGroup grp{};       // CONV
  ConvConfig cfg{};
  // STEM / POOL / UPSAMPLE parameters
  View in, out;
  const float* w27 = nullptr;
  const float* bias = nullptr;
  const void* wpk = nullptr;   // fp16 MFMA / split-f16x3 stem weights
  float stem_scale = 1.f;      // split-f16x3 stem: inverse of the weights' power-of-two scaling
  const void* front_wpk = nullptr;   // the same weights packed for the front stage of model.1 (ConvProblem::front_w)
  float front_scale = 1.f;
  // rows of the output that depend on the frame (Detector::plan_pad_skip), as tile rows per group member; count 0 = all
  int ty_first[kMaxGroup] = {0}, ty_count[kMaxGroup] = {0};
  double flops = 0;      // algorithmic 2*MAC
  double bytes = 0;      // algorithmic: inputs read once + outputs written once + weights
};

// What the C ABI's gtx_detector_* entry points call: one implementation per detector family (gtx_det_config::arch), the way the
// reference swaps YOLO for RTDETR on the model's yaml (geotrax/extract.py:222-225).
class DetectorBase {
 public:
  virtual ~DetectorBase() = default;
  virtual void set_tensor(const std::string& name, const float* data, int ndim, const int64_t* shape) = 0;
  virtual void finalize() = 0;
  virtual void input_size(int* h, int* w) const = 0;
  virtual void detect_dev(const void* frames, int nb, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) = 0;
  virtual void submit_dev(const void* frames, int nb, int h, int w) = 0;
  virtual void collect(int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) = 0;
  virtual void detect_host(const uint8_t* frame, int h, int w, int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) = 0;
  virtual const void* gray(int b, int* gh, int* gw) const = 0;
  virtual void raw_output(int b, float* out, int* n_anchors, bool logits = false) = 0;
  virtual void layer_output(int b, const std::string& layer, float* out, int* h, int* w, int* c) = 0;
  virtual void profile(int nb, int iters, std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms,
                       std::vector<double>& flops, std::vector<double>& bytes) = 0;
  virtual void set_trace(int every_n) = 0;
  virtual void trace_report(std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms, std::vector<double>& flops,
                            std::vector<double>& bytes) = 0;
  virtual void features(int b, float* out, int cap, int* n, int* dim) const = 0;
  virtual bool saturated(bool clear) = 0;
  virtual bool fell_back() const = 0;
  virtual void pad_skip(int* on, int* skipped, int* total) const = 0;
  virtual void sparse_box(int* on, int* overflows) const = 0;
};

class Detector : public DetectorBase {
 public:
  Detector(gtx_ctx* ctx, const gtx_det_config& cfg);
  ~Detector();
  void set_tensor(const std::string& name, const float* data, int ndim, const int64_t* shape) override;
  void finalize() override;
  void input_size(int* h, int* w) const override { *h = lb_.net_h; *w = lb_.net_w; }

  // frames: device pointer, nb frames [h][w][3] u8 back to back. Outputs sized [nb][max_det].
  void detect_dev(const void* frames, int nb, int h, int w, int* n_out, float* xyxy, float* conf,
                  int* cls, float speed_ms[3]) override;
  // asynchronous pair: submit enqueues the whole pass, collect waits for it and unpacks
  void submit_dev(const void* frames, int nb, int h, int w) override;
  void collect(int* n_out, float* xyxy, float* conf, int* cls, float speed_ms[3]) override;
  void detect_host(const uint8_t* frame, int h, int w, int* n_out, float* xyxy, float* conf,
                   int* cls, float speed_ms[3]) override;
  const void* gray(int b, int* gh, int* gw) const override;
  void raw_output(int b, float* out, int* n_anchors, bool logits = false) override;
  void layer_output(int b, const std::string& layer, float* out, int* h, int* w, int* c) override;
  void profile(int nb, int iters, std::vector<std::string>& names, std::vector<int>& launches,
               std::vector<float>& ms, std::vector<double>& flops, std::vector<double>& bytes) override;
  // Live tracing: every `every_n`-th submitted pass gets a HIP event in front of every launch of the
  // forward graph (on the launch stream); collect() folds the elapsed times into per-op totals that
  // trace_report() returns per kernel family and clears. every_n = 0 switches tracing off.
  void set_trace(int every_n) override;
  void trace_report(std::vector<std::string>& names, std::vector<int>& launches, std::vector<float>& ms,
                    std::vector<double>& flops, std::vector<double>& bytes) override;
  int max_det() const { return cfg_.max_det; }
  // gtx_det_config::obj_feats: appearance vectors of image b's boxes of the most recently collected batch, [n][dim] (n = its box count)
  int feat_dim() const { return exact_ ? exact_->feat_dim() : feat_levels_.dim; }
  void features(int b, float* out, int cap, int* n, int* dim) const override;
  // split-f16x3 path: true when some activation of a collected pass (since the last call with clear) had to be clamped to
  // fp16's range on its way into the pair format. collect() then re-runs that batch through an exact-fp32 detector built
  // from the same tensors and every later pass goes there (`ultralytics.half: false` promises fp32's range,
  // default.yaml:245); GTX_SAT_FALLBACK=0 keeps the flag only.
  bool saturated(bool clear) override;
  bool fell_back() const override { return exact_ != nullptr; }
  void pad_skip(int* on, int* skipped, int* total) const override {
    if (on) *on = (!exact_ && pad_skip_on_) ? 1 : 0;
    if (skipped) *skipped = pad_skip_rows_;
    if (total) *total = pad_skip_total_;
  }
  void sparse_box(int* on, int* overflows) const override {
    if (on) *on = (!exact_ && sparse_on_) ? 1 : 0;
    if (overflows) *overflows = sparse_overflows_;
  }

 private:
  void* alloc(size_t bytes);
  View new_view(int h, int w, int c);
  const HostTensor& tensor(const std::string& name) const;
  bool has(const std::string& name) const { return tensors_.count(name) != 0; }
  // graph building
  // up_src: the leading up_src->c channels of x are the 2x nearest upsampling of *up_src and are read from there
  // (split-f16x3 1x1 convs; ConvProblem::in2) -- the slice of x they would occupy is never written
  View conv(const std::string& name, const View& x, int stride, bool act, const View* out_slice,
            const View* residual, const View* up_src = nullptr);
  View c2f(const std::string& pfx, const View& x, bool shortcut, const View* out_slice, const View* up_src = nullptr);
  void build_graph();
  void fuse_front();         // model.1 (3x3 stride 2) + model.2.cv1 (1x1) as one launch on the split-f16x3 path
  void fuse_stem();          // model.0 (the stem) computed inside model.1's launch: its output never reaches HBM
  void run_op(const Op& op, int nb, hipStream_t s);
  void run_forward(int nb, hipStream_t s, bool traced = false);
  void run_post(int nb, hipStream_t s);
  void set_batch(int nb);

  void fall_back_to_exact();
  void release_hidden_layers();     // after the fusions: buffers only the stand-alone forms of fused layers write
  void materialize_hidden_layers(); // ... come back on the first layer_output() that asks for one of them
  struct Hidden { void* token; size_t bytes; void* real; };
  std::vector<Hidden> hidden_;

  gtx_ctx* ctx_;
  gtx_det_config cfg_;
  std::unique_ptr<Detector> exact_;   // the exact-fp32 detector every call is handed to once a split-f16x3 pass has saturated
  int dtype_;                // activation type in HBM (DT_F16 / DT_F32)
  int conv_dtype_;           // what the conv kernels compute in (dtype_, or DT_F32S: split-f16x3 on fp32 activations)
  size_t es_;
  Letterbox lb_{};
  std::map<std::string, HostTensor> tensors_;
  std::vector<DevBuf> bufs_;
  std::vector<Op> ops_;
  std::map<std::string, View> layer_views_;
  bool finalized_ = false;
  int force_kc_ = 0;         // K chunk forced on the convs being built (grouped head stages)
  int force_bn_ = 0;         // cout tile forced on them
  int cur_nb_ = 0;

  View img_;                 // [N][net_h][net_w][4]
  DevBuf frame_stage_;       // device copy of host frames for detect_host
  DevBuf gray_;              // [N][gh][gw] u8
  int gray_h_ = 0, gray_w_ = 0;
  const void* cur_frames_ = nullptr;
  static constexpr int kGrayRing = 16;
  int gray_slot_ = 0, collected_gray_slot_ = 0;
  bool in_flight_ = false;
  int flight_nb_ = 0;

  HeadParams head_{};
  NmsBuffers nms_{};
  DevBuf raw_;               // debug raw output
  int* sat_dev_ = nullptr;   // set by the split convolutions when they clamp (ConvProblem::sat_flag)
  int* h_sat_ = nullptr;     // pinned copy, refreshed by every pass
  bool sat_seen_ = false;
  bool plain_out_ = false;   // convs being built write plain fp32 (head stage 2)
  std::vector<Op> unfused_;  // the stand-alone forms of fused ops (layer_output of an intermediate runs them on demand)
  int* h_out_n_ = nullptr;   // pinned
  FeatLevels feat_levels_{};
  // sparse box branch (head_sparse.hip): on for the split-f16x3 path when the head's layers have the 16x16x32 kernel's weight images
  static constexpr int kSparseCap = 8192;   // candidates per image its buffer holds
  bool sparse_on_ = false;
  SparseBox sparse_{};
  Op head_ops_[3][3];
  std::vector<Op> dense_box_ops_;
  bool dense_head_valid_ = false;
  int* h_count_ = nullptr;   // pinned: candidates per image of the pass in flight
  int sparse_overflows_ = 0;
  void run_dense_box(hipStream_t s);
  // letterbox-padding rows: activations there do not depend on the frame; computed once (finalize), skipped afterwards
  bool pad_skip_on_ = false;
  int pad_skip_rows_ = 0, pad_skip_total_ = 0;   // tile rows skipped / planned over all convolution launches (for the report)
  void plan_pad_skip();
  void prime_pad_skip();
  float* d_feats_ = nullptr;  // [max_batch][max_det][dim]
  float* h_feats_ = nullptr;  // pinned
  std::vector<float> c_feats_;   // the collected batch's vectors
  std::vector<int> c_feat_n_;
  float* h_out_rows_ = nullptr;
  hipEvent_t ev_[4]{};
  hipGraphExec_t graph_exec_ = nullptr;   // captured forward graph for batch size graph_nb_
  int graph_nb_ = 0;
  int trace_every_ = 0, trace_count_ = 0;
  bool flight_traced_ = false;
  std::vector<hipEvent_t> trace_ev_;      // one per op + 1
  std::vector<double> trace_ms_;          // per op
  std::vector<int> trace_n_;
  std::vector<double> trace_flops_, trace_bytes_;   // per op, summed over the traced passes (each at its own batch size)
  hipEvent_t ev_up_[2]{};
};

}  // namespace gtx

struct gtx_detector {
  std::unique_ptr<gtx::DetectorBase> impl;
};
