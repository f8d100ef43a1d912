// Implicit-GEMM convolution for the YOLOv8 Conv block (conv + folded BN + SiLU),
// MI355X / gfx950. Replaces what ultralytics' `Conv.forward_fuse` computes per layer
// (reached from the reference at geotrax/extract.py:153).
//
// Data layout in HBM
//   activations : NHWC, element type T (fp16 or fp32), addressed as
//                 base + ((n*H + y)*W + x)*cstride + coff + c  -- cstride/coff let a conv
//                 read a channel slice of, or write straight into, a concat buffer
//                 (C2f / SPPF / FPN concats never materialise as copies).
//   weights     : pre-packed by pack_conv_weights() into the exact LDS image the kernel
//                 stages per (cout tile, cin chunk): [tap][n][swizzled 16-B chunk].
//   bias        : fp32 [Cout].
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace gtx {

// DT_F32S: fp32-grade activations, 4 bytes per element like DT_F32, stored as (hi, lo) fp16 pairs (split_format.hpp);
// convolutions on the fp16 matrix pipe, three MFMAs per product (conv_igemm_split.hip).
enum DType : int { DT_F16 = 0, DT_F32 = 1, DT_F32S = 2 };
inline size_t dtype_size(int dt) { return dt == DT_F16 ? 2 : 4; }

// One convolution problem. Plain-old-data; copied into kernarg space.
struct ConvProblem {
  const void* in;       // NHWC input
  void* out;            // NHWC output
  const void* wpack;    // packed weights (see pack_conv_weights)
  const float* bias;    // [Cout] fp32 (may be null -> 0)
  const void* res;      // optional residual, same spatial dims as out (may be null)
  int N, H, W;          // input batch / spatial
  int Ho, Wo;           // output spatial
  int Cin, Cout;
  int in_cstride, in_coff;
  int out_cstride, out_coff;
  int res_cstride, res_coff;
  int act;              // 0 = identity, 1 = SiLU, 2 = ReLU (RT-DETR's HGNetv2 blocks); applied before the residual is added
  int tiles_x, tiles_y; // output pixel tiles (TW=16, TH=8); tiles_y counts the tile rows the launch computes
  int ty_first, ty_count; // ty_count > 0: only tile rows [ty_first, ty_first + ty_count) are computed (the others keep what the
                        // buffer holds: Detector's letterbox-padding rows, whose values do not depend on the frame); 0 = all
  int n_ct;             // cout tiles (Cout / BN)
  // Optional second source (split-f16x3 1x1 kernels only): input channels [0, c_split) are the nearest-neighbour 2x
  // upsampling of `in2` ([N][H/2][W/2][in2_cstride], slice at in2_coff) and are read from there at (y/2, x/2); channels
  // [c_split, Cin) come from `in` as usual. This is torch's Upsample + Concat feeding a C2f's first conv without the
  // upsampled copy ever being written (YOLOv8 model.10-12 and model.13-15). c_split = 0: off.
  const void* in2;
  int in2_cstride, in2_coff, c_split;
  int block_begin;      // first logical block of this problem inside a grouped launch
  float acc_scale;      // DT_F32S: inverse of the power of two the packed weights were scaled by (else 1)
  // DT_F32S only (split_format.hpp): in / in2 / res / out are in pair format unless out_plain is set, which keeps the output
  // plain fp32 (the Detect head's last conv stage, read by the decode kernels). sat_flag (may be null): set to 1 when a value
  // had to be clamped to +-65504 on its way into the pair format.
  int out_plain;
  int* sat_flag;
  // DT_F32S, 3x3 stride-2 launches whose workgroup holds every output channel (Cout == the cout tile): a 1x1 convolution
  // fused behind the 3x3 one ("post stage": YOLOv8 model.1 -> model.2.cv1). post_w != null: the tile's SiLU(conv) values are
  // split and staged in LDS as pair rows, multiplied there by the Cout x Cout 1x1 weights (post_w: the packed LDS image of the
  // 1x1 layer, pack_conv_weights_split with 32-channel chunks; post_bias, post_scale = its acc_scale, post_act) and only
  // the 1x1 layer's output is written to `out`: the 3x3 layer's own output never reaches HBM (a launch and 2 x its bytes less).
  const void* post_w;
  const float* post_bias;
  float post_scale;
  int post_act;
  // DT_F32S, single 3x3 stride-2 launches with one cout tile: "front stage" (conv_igemm_split.hip, FrontTile). front_img != null:
  // the layer's input is YOLOv8's stem applied to the RGB0 byte image front_img [N][front_h][front_w_px] (H = front_h / 2,
  // W = front_w_px / 2, Cin = the stem's channels), computed patch by patch inside the workgroup; `in` is not read.
  // front_w: pack_front_weights_split; front_bias: the stem's [Cin] bias; front_scale: its acc_scale.
  const void* front_img;
  const void* front_w;
  const float* front_bias;
  float front_scale;
  int front_h, front_w_px;
};

constexpr int kMaxGroup = 8;
struct ConvGroup {
  ConvProblem p[kMaxGroup];
  int count;
  int total_blocks;
  // Hardware block b runs on XCD b & 7 and is that XCD's block number b >> 3; it takes logical block
  // xcd_begin[b & 7] + (b >> 3) and leaves at once when that is past xcd_begin[(b & 7) + 1]. The ranges are contiguous in
  // member order (an XCD's L2 sees neighbouring tiles of as few members as possible) and hold equal WORK, not equal
  // counts: blocks of a member with a deeper K loop count for more (conv_group_finalize). grid_blocks = 8 x the longest range.
  int xcd_begin[9];
  int grid_blocks;
};

// Kernel family selector; every member of a grouped launch shares one config.
struct ConvConfig {
  int dtype;   // DType
  int ks;      // 1 or 3
  int stride;  // 1 or 2
  int bn;      // cout tile: 32, 64 or 128
  int kc;      // cin elements staged per K chunk
  int variant; // 0 = register-staged kernel (conv_igemm.hip: fp16, exact fp32), 2 = split-f16x3 (conv_igemm_split.hip, DT_F32S), 3 / 4 = its Winograd forms (conv_wino_split.hip), 5 = its 16x16x32 form for 3x3 stride 1 (conv_k32_split.hip), 6 = its 16x16x32 form for plain 1x1 layers (conv_k32p_split.hip)
  int th, tw;  // output pixel tile (rows x cols)
};

// Picks the tile configuration used for a layer shape.
// force_kc / force_bn > 0 pin the K chunk / cout tile (grouped launches: every member must use the same kernel instantiation).
// out_pixels (N x Ho x Wo at the layer's largest batch, 0 = unknown): only read by the GTX_WINO=4 experiment.
ConvConfig conv_pick_config(int dtype, int ks, int stride, int cin, int cout, int force_kc = 0, int force_bn = 0, long out_pixels = 0);

// Host-side weight packing: w is [Cout][KS][KS][Cin] fp32 (OHWI). Returns the packed
// byte image for `cfg` (element type per cfg.dtype). acc_scale (may be null) receives ConvProblem::acc_scale.
std::vector<uint8_t> pack_conv_weights(const float* w_ohwi, int cout, int cin, const ConvConfig& cfg, float* acc_scale = nullptr);

// Fills tiles_x/tiles_y/n_ct/block_begin/total_blocks for a group.
void conv_group_finalize(ConvGroup& g, const ConvConfig& cfg);

// Launch one grouped conv on `stream`.
void conv_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream);

// Algorithmic FLOPs (2*MAC) of a problem.
double conv_flops(const ConvProblem& p, int ks);

// conv_igemm_split.hip
void conv_split_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream);
std::vector<uint8_t> pack_conv_weights_split(const float* w_ohwi, int cout, int cin, const ConvConfig& cfg, float* acc_scale);
std::vector<uint16_t> pack_front_weights_split(const float* w27, int c0, float* acc_scale);

// conv_wino_split.hip: the Winograd F(2x2, 3x3) form of the split-f16x3 3x3 stride-1 convolution (ConvConfig::variant 3)
void conv_wino_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream);
std::vector<uint8_t> pack_conv_weights_wino(const float* w_ohwi, int cout, int cin, const ConvConfig& cfg, float* acc_scale);

// conv_k32_split.hip: the 3x3 stride-1 split-f16x3 convolution on v_mfma_f32_16x16x32_f16 (ConvConfig::variant 5; weights: pack_conv_weights_split, kc = 32)
void conv_k32_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream);

// conv_k32p_split.hip: the 1x1 split-f16x3 convolution on v_mfma_f32_16x16x32_f16 (ConvConfig::variant 6; the weight image of variant 2 with kc = 32)
void conv_k32p_launch(const ConvGroup& g, const ConvConfig& cfg, hipStream_t stream);

// Kernel symbol name as rocprof shows it (for the roofline bookkeeping).
const char* conv_kernel_name(const ConvConfig& cfg);

}  // namespace gtx
