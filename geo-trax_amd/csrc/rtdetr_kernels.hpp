// Non-convolution kernels of the RT-DETR detector (rtdetr.cpp): HGNetv2's stem / depthwise / pool pieces, the token-side
// arithmetic of AIFI and the decoder (fp32 MFMA linear layers, LayerNorm, multi-head attention, multi-scale deformable
// attention sampling), query selection and the final score / box stage. gfx950 only. Stands in for ultralytics'
// nn.modules {HGStem, DWConv, AIFI, RTDETRDecoder, MSDeformAttn} + RTDETRPredictor.postprocess underneath
// RTDETR(model).track() (geotrax/extract.py:222-225, :153). Host launch wrappers; kernels live in rtdetr_kernels.hip.
//
// Tensors come in two kinds. "Map" tensors are NHWC activations in the detector's activation format `fmt` (DT_F16 /
// DT_F32 / DT_F32S = the pair format of split_format.hpp), addressed through (cstride, coff) like every convolution's.
// "Token" tensors are plain fp32 rows [M][ld] (M = images x tokens): AIFI's internals and everything the decoder does on its
// 300 queries runs at exact fp32 on v_mfma_f32_16x16x4_f32.
#pragma once
#include "common.hpp"
#include "conv_igemm.hpp"

namespace gtx {

struct RtMap {            // one NHWC map tensor (or a channel slice of one)
  void* ptr;
  int h, w, cstride, coff, c;
};

// HGStem.stem1: Conv(3, c0, 3, 2) + ReLU on the RGB0 byte image [N][H][W][4] (x / 255 applied per byte); w27 [27][c0]
// tap-major ((ky * 3 + kx) * 3 + channel), bias [c0]; out [N][H/2][W/2] map, c0 % 8 == 0.
void launch_rt_stem1(int fmt, const void* img, int n, int H, int W, const float* w27, const float* bias, const RtMap& out, int* sat, hipStream_t s);
// MaxPool2d(2, 1, ceil_mode) on F.pad(x, [0, 1, 0, 1]): out(y, x) = max over the 2 x 2 window at (y, x), zeros past the border
void launch_rt_pool2(int fmt, const RtMap& in, const RtMap& out, int n, int* sat, hipStream_t s);
// depthwise k x k (3 or 5), stride 1 or 2, pad k / 2; w [k * k][C] tap-major, bias [C]; act: 0 none, 2 ReLU
void launch_rt_dwconv(int fmt, const RtMap& in, const RtMap& out, int n, int k, int stride, const float* w, const float* bias, int act, int* sat, hipStream_t s);
// nearest 2x upsampling into a channel slice (raw copy of 8-channel groups)
void launch_rt_upsample2x(int fmt, const RtMap& in, const RtMap& out, int n, hipStream_t s);
// map -> tokens: src [N * h * w][C] plain fp32 and, when q != null, q = src + pos (pos [h * w][C])
void launch_rt_tokens_in(int fmt, const RtMap& in, int n, const float* pos, float* src, float* q, hipStream_t s);
// zero the rows of anchors outside (eps, 1 - eps) (RTDETRDecoder._generate_anchors' valid_mask) of a level's map
void launch_rt_mask_invalid(int fmt, const RtMap& m, int n, int level, hipStream_t s);

// Y[M][ldy] (+coly) = act(X[M][K] (+ X2 for the first x2_cols columns) . W[Nout][K]^T + bias) (+ R); K % 16 == 0, Nout % 16 == 0. act: 0 none, 2 ReLU, 3 GELU (erf)
struct RtLinear {
  const float* x; int ldx;
  const float* x2; int ldx2;      // optional second addend of the input rows (query + query_pos) ...
  int x2_cols;                    // ... for the output columns [0, x2_cols) only: q and k of an in_proj take query + pos, v the query alone
  const float* w; const float* bias;
  const float* res; int ldr;      // optional residual added after the activation
  float* y; int ldy;
  int M, K, Nout, act;
};
void launch_rt_linear(const RtLinear& p, hipStream_t s);

// LayerNorm over the last dim (C % 8 == 0, C <= 1024), eps 1e-5. Rows in: tokens (in_fmt = DT_F32, cstride = C) or a map.
struct RtRows { void* ptr; int cstride, coff; int fmt; };
void launch_rt_layernorm(const RtRows& in, const RtRows& out, long rows, int C, const float* gamma, const float* beta, int* sat, hipStream_t s);

// softmax(Q K^T / sqrt(d)) V per image and head on token rows: qkv [N * T][ld] with q at column 0, k at C, v at 2C
void launch_rt_mha(const float* qkv, int ld, int n, int T, int C, int heads, float* out, int ldo, hipStream_t s);

// query selection: per image, the nq anchors with the largest max-over-classes score, descending (ties: lower anchor index
// first). scores: per level maps [N][h][w][cs] with the classes in channels [0, nc): plain fp32 (fmt DT_F32, both fp32-grade
// paths) or fp16 (fmt DT_F16, `half: true`).
struct RtLevels {
  const void* ptr[3];
  int h[3], w[3], cstride[3], coff[3];
  int n_levels;
};
void launch_rt_topk(int fmt, const RtLevels& scores, int nc, int n, int nq, unsigned* keys_scratch, int* out_idx, hipStream_t s);
// embed[n][q][:] = enc(level, y, x)[:] of the selected anchors (map format -> plain), anchor logits [N * nq][4]
void launch_rt_gather(int fmt, const RtLevels& enc, int C, int n, int nq, const int* idx, float* embed, float* anchors, hipStream_t s);
// reference boxes: mode 0: refer = sigmoid(delta + anchors); mode 1: refer = sigmoid(delta + inverse_sigmoid(refer)).
// delta [M][ldd] (first 4 columns), refer [M][16] (columns 4.. stay zero: the K = 16 input of query_pos_head)
void launch_rt_refer(const float* delta, int ldd, const float* anchors, float* refer, int M, int mode, hipStream_t s);
// multi_scale_deformable_attn_pytorch: offaw [M][nh * L * P * 3] (sampling_offsets then attention_weights), refer [M][16],
// value levels (map format, the layer's hd channels at coff), out [M][hd] plain
void launch_rt_deform(int fmt, const RtLevels& value, int hd, int nh, int npts, const float* offaw, const float* refer, int n, int nq, float* out, hipStream_t s);
// RTDETRPredictor.postprocess: sigmoid, max / argmax over classes, conf + class filter, descending score, boxes x frame size.
// logits [N * nq][ldl], refer [N * nq][16]. out_rows [N][max_det][6], out_n [N], raw [N][nq][4 + nc] (xywh + scores; may be null)
void launch_rt_post(const float* logits, int ldl, const float* refer, int n, int nq, int nc, float conf, const unsigned long long class_mask[2],
                    int frame_w, int frame_h, int max_det, float* out_rows, int* out_n, float* raw, hipStream_t s);

}  // namespace gtx
