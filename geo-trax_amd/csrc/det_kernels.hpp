// Non-GEMM kernels of the detector: letterbox/normalise (+gray), the 3-channel stem conv,
// SPPF pooling, nearest upsample, head decode (final 1x1 convs + DFL + sigmoid + threshold +
// compaction) and NMS. gfx950 only. Host launch wrappers; kernels live in det_kernels.hip.
#pragma once
#include <vector>

#include "common.hpp"
#include "conv_igemm.hpp"

namespace gtx {

// ---- letterbox geometry (ultralytics LetterBox.__call__, reached from extract.py:153) ----
struct Letterbox {
  int src_h, src_w;    // frame
  int net_h, net_w;    // network input
  int new_h, new_w;    // resized (unpadded) content
  int top, left;       // padding
  double gain;         // min(net_h/src_h, net_w/src_w) of the *square* request
};
Letterbox letterbox_geometry(int src_h, int src_w, int imgsz, bool rect, int stride);

// frame: BGR u8 [N][h][w][3] device; img: [N][net_h][net_w][4] T (RGB0, /255, pad 114/255);
// gray: u8 [N][gh][gw] or null (full-res BGR2GRAY then 2x2 mean; requires gh=h/2, gw=w/2).
void launch_preprocess(int dtype, const uint8_t* frames, int n, const Letterbox& lb, void* img,
                       uint8_t* gray, int gh, int gw, hipStream_t s);

// Stem: Conv(3, c0, k=3, s=2) + bias + SiLU on the RGB0 image. w: [27][c0] fp32 (tap-major:
// (ky*3+kx)*3 + c), bias [c0]. c0 must be a multiple of 16 and <= 64.
// wpk_f16: weights packed by pack_stem_weights_f16 (fp16 MFMA path) or null (VALU path).
void launch_stem(int dtype, const void* img, int n, int h, int w, const float* w27, const float* bias,
                 const void* wpk_f16, int c0, void* out, int ho, int wo, hipStream_t s, float acc_scale = 1.f);
std::vector<uint16_t> pack_stem_weights_f16(const float* w27, int c0);
// dtype DT_F32S (fp32 image and output, split-f16x3 MFMA): hi + lo packed weights, *acc_scale = inverse of their power-of-two scaling
std::vector<uint16_t> pack_stem_weights_split(const float* w27, int c0, float* acc_scale);

// SPPF pools: channels [0,c) -> 5x5 / 9x9 / 13x13 clipped-window maxima at [c,2c) [2c,3c) [3c,4c).
void launch_sppf_pool(int dtype, void* x, int n, int h, int w, int c, hipStream_t s);

void launch_upsample2x(int dtype, const void* x, int n, int h, int w, int c, int in_cstride,
                       int in_coff, void* y, int out_cstride, int out_coff, hipStream_t s);

// ---- head decode + NMS ----
constexpr int kMaxLevels = 3;
struct HeadLevel {
  const void* feat;     // [N][h][w][cstride] T: channels [0,cb) box branch, [cb, cb+cc) cls branch
  int h, w, cstride;
  int cb, cc;           // box / cls feature channels (64 / 128 for YOLOv8s)
  const float* wb;      // [cb][64] fp32  final box 1x1 conv (4*reg_max outputs), transposed
  const float* bb;      // [64]
  const float* wc;      // [nc][cc] fp32  final cls 1x1 conv
  const float* bc;      // [nc]
  float stride;         // 8 / 16 / 32
  int anchor_begin;     // index of this level's first anchor
};
struct HeadParams {
  HeadLevel lv[kMaxLevels];
  int n_levels;
  int n_anchors;        // per image
  int nc;
  float conf;           // score threshold (strict >)
  unsigned long long class_mask[2];  // bit c set = class c kept (ultralytics `classes`); 128 classes
};

struct NmsBuffers {
  // per image slot b: candidates are stored at [b*cap ...]
  int cap;              // candidate capacity per image
  int* count;           // [N] number of candidates (atomic)
  float* cand_score;    // [N][cap]
  int* cand_anchor;     // [N][cap]
  int* cand_cls;        // [N][cap]
  float* cand_box;      // [N][cap][4] xyxy network pixels
  // sorted (score desc, anchor asc), truncated to nms_cap
  int nms_cap;
  int* sorted_n;        // [N]
  float* s_box;         // [N][nms_cap][4]
  float* s_score;       // [N][nms_cap]
  int* s_cls;           // [N][nms_cap]
  int* s_anchor;        // [N][nms_cap] (may be null: anchors of the kept boxes not wanted)
  unsigned long long* mask;  // [N][nms_cap][nms_cap/64]
  // final
  int max_det;
  int* out_n;           // [N]
  float* out_rows;      // [N][max_det][6] x1 y1 x2 y2 conf cls (frame pixels)
  int* out_anchor;      // [N][max_det] anchor index of every output row (may be null)
  // ... and the candidates filed by level for it (indices into the candidate arrays): lvl_count [N][kMaxLevels],
  // lvl_list [N][kMaxLevels][lvl_cap]; null: not kept
  int* lvl_count;
  int* lvl_list;
  int lvl_cap;
};

// Per-object appearance vectors "from the detector" (ultralytics BoT-SORT `with_reid: true, model: auto`,
// default.yaml:376-379): the Detect layer's three input maps, every level's channels averaged in consecutive groups down to
// the narrowest level's width (`dim`), read at the anchor each kept box came from (predictor.get_obj_feats).
struct FeatLevels {
  const void* feat[kMaxLevels];   // [N][h][w][cstride] activations of the detector's dtype (pair format for DT_F32S)
  int h[kMaxLevels], w[kMaxLevels], cstride[kMaxLevels], coff[kMaxLevels], c[kMaxLevels];
  int anchor_begin[kMaxLevels];
  int n_levels;
  int dim;                        // min over levels of c
};
// out: [N][max_det][dim] fp32, rows [0, out_n[n]) of image n written
void launch_obj_feats(int dtype, const FeatLevels& fl, int n, const NmsBuffers& nb, float* out, hipStream_t s);

// launch_head_candidates = the score gate (class branch, compaction) followed by the box decode of the candidates; the two halves
// are also available on their own so that the sparse box branch (head_sparse.hip) can run between them.
void launch_head_candidates(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s);
void launch_head_gate(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s);
void launch_head_boxes(int dtype, const HeadParams& hp, int n, const NmsBuffers& nb, hipStream_t s);

// The Detect box branch (cv2[l][0]: Cin -> 64 and cv2[l][1]: 64 -> 64, both 3x3 + SiLU) evaluated at the candidate anchors only
// (head_sparse.hip; split-f16x3 path). Per level: the Detect input (pair format), the packed weight image of the first layer's
// 64-cout tile and of the second layer (pack_conv_weights_split with 32-channel chunks), their biases and power-of-two scales.
struct SparseBoxLevel {
  const void* in;
  int H, W, cstride, coff, cin;
  const void* w1; const float* b1; float sc1;
  const void* w2; const float* b2; float sc2;
  int anchor_begin;
  const float* wb; const float* bb; float stride;   // the level's final box 1x1 convolution ([64][64] transposed, bias) and its stride: HeadLevel
};
struct SparseBox {
  SparseBoxLevel lv[kMaxLevels];
  int n_levels;
  int cap;              // candidates per image the buffer holds (more: the caller runs the dense layers)
  int* sat_flag;
};
void launch_head_sparse_box(const SparseBox& sb, int n, const NmsBuffers& nb, hipStream_t s);
// Decode every anchor (debug / parity): out [N][A][4+nc] fp32 = xywh (network px) + class scores
// (or class logits).
void launch_head_raw(int dtype, const HeadParams& hp, int n, float* out, bool logits, hipStream_t s);
void launch_nms(const NmsBuffers& nb, int n, float iou_thr, bool agnostic, int max_nms,
                const Letterbox& lb, hipStream_t s, int which = 0);
// true when the single-workgroup kernel handles an image with that many candidates (the general kernels then have nothing to do)
bool nms_small_covers(int candidates, int max_det);

}  // namespace gtx
