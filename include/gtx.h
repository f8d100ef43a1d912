/*
 * gtx.h -- C ABI of libgtx.so, the MI355X (gfx950) implementation of the geo-trax
 * per-frame extraction hot path.
 *
 * The reference (rfonod/geo-trax) has no FFI of its own: its hot path is two duck-typed
 * Python objects called from one loop, geotrax/extract.py:134-214 --
 *   model.track(frame, **cfg, persist=True)                      extract.py:153
 *   Stabilizer(**cfg).set_ref_frame / .stabilize /
 *     .transform_cur_boxes / .get_cur_trans_matrix               extract.py:139,177-184
 * plus cv2.perspectiveTransform in geotrax/georeference.py:599-605 and
 * cv2.warpPerspective in geotrax/visualize.py:289.
 * Each entry point below names the reference call it stands in for. The Python classes in
 * geo-trax_amd/geotrax_amd/ (YOLO, Stabilizer) bind these symbols with ctypes and keep the
 * reference's method names, argument meaning and error behaviour.
 *
 * Conventions
 *   - every function returns 0 on success or a negative gtx_status; the text of the last
 *     error on the calling thread is available from gtx_last_error();
 *   - no C++ exception crosses this boundary;
 *   - the library owns all device memory; the caller owns every host pointer it passes and
 *     may reuse it as soon as the call returns (calls are synchronous unless named *_submit);
 *   - a context (and everything created from it) is bound to one GPU and is not thread-safe:
 *     one context = one host thread, exactly like the reference's single-threaded loop;
 *   - plain C types only: pointers, sizes, ints, floats, doubles.
 */
#ifndef GTX_H_
#define GTX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: gtx_det_config.fp32_split, gtx_tracker_config.{delta_t, inertia, use_byte, min_hits}, gtx_stab_config.clahe appended;
 *    gtx_tracker_replay, gtx_op_clahe, gtx_warp_frame_dev, gtx_yuv420_to_bgr_dev, gtx_stabilizer_{pattern, last_ms} added.
 *    A binder checks gtx_abi_version() against the header it was written for before passing any struct.
 * 3: gtx_stab_config.{affine, filter_type} appended; gtx_tracker_config.type 3 (deepocsort).
 * 4: gtx_op_linear_assignment and gtx_detector_saturated added; GTX_F32S activations live in HBM as (hi, lo) fp16 pairs
 *    (host arrays handed to gtx_op_* stay plain fp32).
 * 5: gtx_feeder_* (read-ahead frame source) added; a saturating split-f16x3 pass is
 *    re-run by the detector through the exact-fp32 kernels (gtx_detector_saturated reports that it happened).
 * 6: gtx_device_open_null_stream, gtx_write_table_f32 / _f64, gtx_write_csv and gtx_track_anchor_walk added.
 * 7: gtx_streams_overlap, gtx_device_mem_info and gtx_sift_stage_ms added; gtx_tracker_config.type 4 (fasttrack) with its parameters appended to the struct.
 * 8: the appearance branch of BoT-SORT on detector-derived vectors (`with_reid: true, model: auto`): gtx_det_config.obj_feats,
 *    gtx_tracker_config.{with_reid, proximity_thresh, appearance_thresh} appended; gtx_detector_features, gtx_tracker_update_feats added;
 *    gtx_tracker_config.type 5 (tracktrack) with its parameters appended; gtx_detector_sparse_box and gtx_detector_pad_skip added.
 * 9: gtx_det_config.arch appended: 1 = RT-DETR (the reference swaps YOLO for RTDETR on the model's yaml, extract.py:222-225);
 *    gtx_tracker_config.alpha_fixed_emb appended, with_reid also read by types 3 (deepocsort) and 5 (tracktrack);
 *    gtx_op_estimate_affine_partial added (GMC methods orb / sift).
 * 10: gtx_ecc_* added (GMC method ecc). */
#define GTX_ABI_VERSION 10

typedef enum gtx_status {
  GTX_OK = 0,
  GTX_ERR_INVALID = -1,     /* bad argument */
  GTX_ERR_HIP = -2,         /* HIP runtime error */
  GTX_ERR_UNSUPPORTED = -3, /* shape / option outside what the kernels implement */
  GTX_ERR_STATE = -4,       /* call order (e.g. stabilize before set_ref_frame) */
  GTX_ERR_INTERNAL = -5
} gtx_status;

/* GTX_F32S (gtx_op_conv2d only): fp32 arrays like GTX_F32, the convolution runs as split-f16x3 (hi + lo fp16
 * operands, three fp16 MFMAs per product, fp32 accumulate) -- what a detector with fp32_split = 1 uses. */
typedef enum gtx_dtype { GTX_F16 = 0, GTX_F32 = 1, GTX_F32S = 2 } gtx_dtype;

typedef struct gtx_ctx gtx_ctx;
typedef struct gtx_detector gtx_detector;
typedef struct gtx_stabilizer gtx_stabilizer;
typedef struct gtx_tracker gtx_tracker;

/* ------------------------------------------------------------------ library / context */

int gtx_abi_version(void);
/* Last error message of the calling thread ("" if none). Never NULL. */
const char* gtx_last_error(void);
/* Number of visible HIP devices (0 if none / runtime unavailable). */
int gtx_device_count(void);

/* One context per GPU. Replaces the implicit torch device selection the reference leaves to
 * ultralytics (cfg ultralytics.device, geotrax/cfg/default.yaml:236). */
int gtx_ctx_create(int device, gtx_ctx** out);
/* Same, the context's stream gets the device's highest priority when high_priority != 0: for the
 * short kernels of a latency-critical consumer (the stabilizer) running beside the detector. */
int gtx_ctx_create_prio(int device, int high_priority, gtx_ctx** out);
void gtx_ctx_destroy(gtx_ctx* ctx);
int gtx_ctx_synchronize(gtx_ctx* ctx);
/* Makes the device's null stream exist now (one 4-byte fill on it, waited for). The HIP runtime gives every stream its
 * place on one of GPU_MAX_HW_QUEUES (4) hardware queues when the stream is created -- the first four open a queue each,
 * a later one joins the queue that carries the fewest streams (ties: the highest-numbered queue) -- and the null stream
 * takes its place the first time anything synchronous (hipMemcpy, hipMemset: the library's set-up paths) runs. A caller
 * that lays out several contexts for concurrency (geotrax_amd/engine.py StreamPlan) calls this at a fixed point of its
 * creation order, so that which streams share a queue does not depend on when the first set-up copy happens. Idempotent. */
int gtx_device_open_null_stream(int device);
/* hipMemGetInfo of the device: what registration at the reference's size (a 15 000-px orthophoto: ~55 GB of pyramids) reports as its
 * footprint. */
int gtx_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* Do kernels on the two contexts' streams run at the same time? One idle wave spins for spin_us microseconds on a's stream
 * (ms_single: host-timed, best of three) and then on both streams at once (ms_pair). Streams that share a hardware queue run
 * in order: ms_pair ~ 2 x ms_single; streams on queues of their own: ms_pair ~ ms_single. The extract engine's stream plan
 * (geotrax_amd/engine.py) is computed from a measured rule of the HIP runtime, not a documented one; this is its self-check. */
int gtx_streams_overlap(gtx_ctx* a, gtx_ctx* b, float spin_us, float* ms_single, float* ms_pair);
/* Raw device memory for callers that keep inputs resident in HBM (bench.py). */
int gtx_dev_alloc(gtx_ctx* ctx, size_t bytes, void** dptr);
int gtx_dev_free(gtx_ctx* ctx, void* dptr);
int gtx_dev_upload(gtx_ctx* ctx, void* dptr, const void* host, size_t bytes);
int gtx_dev_download(gtx_ctx* ctx, void* host, const void* dptr, size_t bytes);

/* One planar YUV 4:2:0 (I420) frame in HBM -> packed BGR u8 in HBM: the colour conversion a video decoder applies
 * before extract.py:146 sees the frame (cv2.VideoCapture.read() returns BGR). BT.601 limited range, the fixed-point
 * constants of cv2.cvtColor(COLOR_YUV2BGR_I420), one chroma sample per 2x2 luma block. yuv_dptr: h*w luma bytes, then
 * the U and V planes of ((h+1)/2)*((w+1)/2) bytes each; bgr_dptr: h*w*3 bytes. Enqueued on the context's stream. */
int gtx_yuv420_to_bgr_dev(gtx_ctx* ctx, const void* yuv_dptr, int h, int w, void* bgr_dptr);

/* Read-ahead frame source: the `cap.read()` at the top of the reference's loop (geotrax/extract.py:146) taken off
 * the thread that drives the detector. Frames of an uncompressed file (.y4m payloads, the data block of a .npy) are
 * read by the feeder's own threads with pread() straight into a ring of pinned host slots, copied to a ring of
 * device batches on the feeder's own stream (I420 frames converted to BGR there, as gtx_yuv420_to_bgr_dev does) and
 * handed out in clip order as device pointers of `batch` contiguous BGR frames.
 *   kind: 0 = frames are BGR u8 [h][w][3]; 1 = I420 planes (h*w + 2*((h+1)/2)*((w+1)/2) bytes).
 *   ring: device batches (and pinned slots x batch) the feeder owns; it reads ahead until all of them are full.
 * gtx_feeder_open_file: deliver the n_frames frames whose payloads start at offsets[i] (delivery order = array order),
 *   read by n_threads reader threads. gtx_feeder_open_push / _push / _finish: the caller's thread supplies host frames
 *   (any source without a flat file layout); push blocks while the ring is full, finish marks the end of the source.
 * gtx_feeder_next: blocks until the next batch's copies are enqueued; *n = frames in it (batch, fewer for the last one,
 *   0 at the end of the source: *dptr = NULL). A read error surfaces here as a negative status after the batches that
 *   preceded it were delivered. gtx_feeder_wait: the consumer context's stream waits for batch `batch_index` to be
 *   resident (no host thread blocks); consumer NULL: the calling thread waits. gtx_feeder_release: the first n_batches
 *   batches have been consumed (the kernels reading them are complete); their slots are read into again.
 * Thread safety: next / wait / release from one consumer thread, push / finish from one producer thread (push_at: any). */
typedef struct gtx_feeder gtx_feeder;
int gtx_feeder_create(int device, int h, int w, int kind, int batch, int ring, gtx_feeder** out);
/* Same, the transfers and conversions run on copy_ctx's stream instead of a stream of the feeder's own (the context must
 * outlive the feeder and be used for nothing else meanwhile). HIP deals streams to a few hardware queues in creation order and
 * streams that share a queue run in order: a caller that creates its streams in a deliberate order (geotrax_amd.engine does)
 * decides this way whose launches the transfers may delay. */
int gtx_feeder_create_on(gtx_ctx* copy_ctx, int h, int w, int kind, int batch, int ring, gtx_feeder** out);
void gtx_feeder_destroy(gtx_feeder* f);
int gtx_feeder_open_file(gtx_feeder* f, const char* path, const int64_t* offsets, int64_t n_frames, int n_threads);
/* The frames are host arrays (decoded footage held in memory, an ndarray [F][h][w][3]): frames[i] points at frame i's bytes,
 * delivery order = array order; the library's reader threads copy them into the pinned ring. The arrays must stay alive and
 * unchanged until the feeder is destroyed. */
int gtx_feeder_open_memory(gtx_feeder* f, const void* const* frames, int64_t n_frames, int n_threads);
int gtx_feeder_open_push(gtx_feeder* f);
int gtx_feeder_push(gtx_feeder* f, const void* frame, size_t bytes);
/* Push mode with several producer threads: frame number i of the source (each number exactly once, any order, any thread;
 * a push blocks while frame i's batch is more than `ring` batches ahead of the consumer). gtx_feeder_finish after the last
 * push has returned. */
int gtx_feeder_push_at(gtx_feeder* f, int64_t i, const void* frame, size_t bytes);
int gtx_feeder_finish(gtx_feeder* f);
/* Abandons the source: worker threads end, a blocked gtx_feeder_push / gtx_feeder_next returns with an error. Call it
 * (and join the pushing thread) before gtx_feeder_destroy when the run is given up half way. */
int gtx_feeder_stop(gtx_feeder* f);
int gtx_feeder_next(gtx_feeder* f, void** dptr, int* n, int64_t* batch_index);
int gtx_feeder_wait(gtx_feeder* f, int64_t batch_index, gtx_ctx* consumer);
int gtx_feeder_release(gtx_feeder* f, int64_t n_batches);

/* ------------------------------------------------------------------ operator level
 * Single operators of the detector, exposed so the parity tests can check every kernel
 * against oracle/ on the exact layer shapes. Host buffers in, host buffers out. */

typedef struct gtx_conv_desc {
  int dtype;               /* gtx_dtype of activations (weights/bias always given as fp32) */
  int n, h, w;             /* input batch / height / width */
  int cin, cout;
  int ksize;               /* 1 or 3 (padding = ksize/2) */
  int stride;              /* 1 or 2 */
  int act;                 /* 1 = SiLU, 0 = identity */
  int in_cstride, in_coff; /* input buffer has in_cstride channels per pixel; conv reads
                              channels [in_coff, in_coff+cin) */
  int out_cstride, out_coff;
  int has_residual;        /* y += residual (residual laid out like the output slice) */
} gtx_conv_desc;

/* ultralytics Conv.forward_fuse: act(conv2d(x, w) + b) on NHWC data.
 * x: [n,h,w,in_cstride] dtype; w_ohwi: [cout,k,k,cin] fp32; bias: [cout] fp32 or NULL;
 * residual: [n,ho,wo,cout] dtype or NULL; y: [n,ho,wo,out_cstride] dtype (only the slice is
 * written; the rest of y is copied through from the caller's buffer). */
int gtx_op_conv2d(gtx_ctx* ctx, const gtx_conv_desc* d, const void* x, const float* w_ohwi,
                  const float* bias, const void* residual, void* y);
/* Repeats the same launch `iters` times and returns the mean kernel time (ms) measured with
 * HIP events on the launch stream, plus the algorithmic FLOPs of one launch. */
int gtx_op_conv2d_time(gtx_ctx* ctx, const gtx_conv_desc* d, int iters, float* ms_per_launch,
                       double* flops);
/* Host only (no GPU call): how a grouped convolution launch is cut over the 8 XCDs. n_members problems of
 * blocks[i] workgroups with cin[i] input channels each, in launch order. xcd_begin[0..8]: hardware block b takes
 * logical block xcd_begin[b & 7] + (b >> 3) and exits when that reaches xcd_begin[(b & 7) + 1]; the ranges hold equal
 * work (blocks weighted by their K depth), not equal counts. grid_blocks = 8 x the longest range. */
int gtx_op_conv_xcd_ranges(int n_members, const int* blocks, const int* cin, int xcd_begin[9], int* grid_blocks);

/* SPPF max-pool cascade (three 5x5/s1/p2 pools of ultralytics SPPF.forward): reads channels
 * [0,c) of x and writes the 5x5, 9x9 and 13x13 window maxima to channels [c,2c), [2c,3c),
 * [3c,4c) of the same NHWC buffer (cstride = 4c). */
int gtx_op_sppf_pool(gtx_ctx* ctx, int dtype, int n, int h, int w, int c, void* x_inout);
/* nn.Upsample(scale_factor=2, mode="nearest") writing into a channel slice. */
int gtx_op_upsample2x(gtx_ctx* ctx, int dtype, int n, int h, int w, int c, const void* x,
                      int in_cstride, int in_coff, void* y, int out_cstride, int out_coff);

/* Brute-force L2 2-nearest-neighbour search of unit-norm 128-d float descriptors (RootSIFT): what
 * cv2.BFMatcher(NORM_L2).knnMatch(query, train, k=2) returns inside stabilo for the orthophoto
 * registration (geotrax/utils/registration.py:59-85, matcher_name='bf'). query [nq][128], train
 * [nt][128] fp32 host arrays; idx1/idx2 = nearest / second nearest train row (-1 if absent), d1/d2
 * their L2 distances (exact fp32; the search itself runs on fp16 MFMA). iters > 0 also times the
 * device passes (ms_per_pass, for bench/roofline). */
int gtx_op_match_2nn(gtx_ctx* ctx, const float* query, int nq, const float* train, int nt,
                     int* idx1, int* idx2, float* d1, float* d2, int iters, float* ms_per_pass);

/* LetterBox + BGR->RGB + /255 (ultralytics predictor preprocess, reached from
 * extract.py:153) fused with the stabilizer's gray + resize (stabilo, extract.py:177,181).
 * frame: BGR u8 [h,w,3]. out_img: [net_h,net_w,4] dtype (RGB0). out_gray: u8
 * [gray_h,gray_w] (may be NULL). */
int gtx_op_preprocess(gtx_ctx* ctx, int dtype, const uint8_t* frame, int h, int w, int net_h,
                      int net_w, void* out_img, uint8_t* out_gray, int gray_h, int gray_w);

/* ------------------------------------------------------------------ detector
 * Stands in for ultralytics YOLO(model).predict half of model.track() -- preprocess,
 * YOLOv8 forward, decode, NMS, scale back to frame coordinates (extract.py:153,222). */

typedef struct gtx_det_config {
  int imgsz;        /* cfg ultralytics.imgsz (default.yaml:235) */
  float conf;       /* ultralytics.conf  */
  float iou;        /* ultralytics.iou   */
  int max_det;      /* ultralytics.max_det */
  int agnostic_nms; /* ultralytics.agnostic_nms */
  int half;         /* ultralytics.half: 0 = fp32 activations/MFMA, 1 = fp16 */
  int rect;         /* ultralytics.rect: 0 = pad to imgsz x imgsz, 1 = minimal stride-32 rectangle */
  int nc;           /* number of classes of the model */
  int n_classes;    /* length of classes[]; 0 = keep all (ultralytics.classes) */
  int classes[80];
  int max_batch;    /* frames per forward pass the buffers are sized for (>=1) */
  int frame_h, frame_w; /* source frame size the buffers are sized for */
  int fp32_split;   /* half == 0 only. 0: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32). 1: "split-f16x3" -- fp32
                     * activations in HBM, every conv operand split into hi + lo fp16 parts in LDS, three fp16
                     * MFMAs per product with fp32 accumulation (22 significand bits per operand; same
                     * detections as the exact path within the fp32 tolerance, ~3/16 of its matrix cost) */
  int obj_feats;    /* 1: keep an appearance vector per output box (gtx_detector_features) -- what ultralytics hands BoT-SORT under
                     * `with_reid: true, model: auto` (default.yaml:376-379; engine/predictor.py get_obj_feats): the Detect layer's
                     * three input maps, each level's channels averaged in consecutive groups down to the narrowest level's width
                     * (128 for YOLOv8s), read at the anchor the box came from */
  int arch;         /* 0: YOLOv8 (Detect head + NMS). 1: RT-DETR (rtdetr-l topology: HGNetv2, AIFI + CCFM, deformable-attention
                     * decoder, no NMS; ultralytics RTDETR, extract.py:222-225). The frame is then stretched to imgsz x imgsz
                     * (RTDETRPredictor.pre_transform: scale_fill), iou / agnostic_nms / rect are not read, obj_feats must be 0; half = 1: fp16 maps and
                     * weights on the fp16 MFMA convolutions, the token side (AIFI, the decoder's queries) stays fp32;
                     * gtx_detector_raw_output returns [queries][4 + nc] = xywh normalised to the frame + class scores */
} gtx_det_config;

int gtx_detector_create(gtx_ctx* ctx, const gtx_det_config* cfg, gtx_detector** out);
void gtx_detector_destroy(gtx_detector* det);
/* Weight hand-over, one tensor at a time, using ultralytics state_dict names of the *fused*
 * model (e.g. "model.0.conv.weight" OIHW fp32, "model.0.conv.bias"). The Python host reads the
 * safetensors file. Replaces YOLO(model=...) (extract.py:222). */
int gtx_detector_set_tensor(gtx_detector* det, const char* name, const float* data, int ndim,
                            const int64_t* shape);
/* Packs weights for the kernels, plans buffers, captures the hipGraph. */
int gtx_detector_finalize(gtx_detector* det);
/* Network input size actually used (after imgsz / rect / stride rounding). */
int gtx_detector_input_size(gtx_detector* det, int* net_h, int* net_w);

/* One frame, host BGR u8 [h,w,3] -> boxes in frame pixels, sorted by confidence (the order
 * ultralytics' NMS returns). Output arrays must hold max_det entries.
 * speed_ms[3] = preprocess, inference, postprocess -- the `results[0].speed` dict that
 * extract.py:155-156 sums. */
int gtx_detector_detect(gtx_detector* det, const uint8_t* frame_bgr, int h, int w, int* n_out,
                        float* xyxy, float* conf, int* cls, float speed_ms[3]);
/* Same, frame already resident in HBM (dptr from gtx_dev_alloc). */
int gtx_detector_detect_dev(gtx_detector* det, const void* frame_dptr, int h, int w, int* n_out,
                            float* xyxy, float* conf, int* cls, float speed_ms[3]);
/* Batched variant: nb frames resident in HBM back to back; outputs are [nb][max_det]. */
int gtx_detector_detect_batch_dev(gtx_detector* det, const void* frames_dptr, int nb, int h, int w,
                                  int* n_out, float* xyxy, float* conf, int* cls,
                                  float speed_ms[3]);
/* Asynchronous pair for pipelining: _submit_dev enqueues the whole pass for nb frames resident in
 * HBM on the context's stream and returns; _collect waits for that batch and fills the outputs
 * ([nb][max_det] arrays). One batch may be in flight per detector. While it runs the caller can
 * drive the tracker and the stabilizer (another context / stream) on the previous batch. */
int gtx_detector_submit_dev(gtx_detector* det, const void* frames_dptr, int nb, int h, int w);
int gtx_detector_collect(gtx_detector* det, int* n_out, float* xyxy, float* conf, int* cls,
                         float speed_ms[3]);
/* Device pointer of the half-resolution gray image the preprocess pass wrote for batch slot b of
 * the most recently *collected* batch (the images live in a 32-deep ring: an image stays valid until
 * fourteen more batches have been submitted after the one that follows it), or NULL. The stabilizer consumes it so the frame is read from HBM once. */
const void* gtx_detector_gray(gtx_detector* det, int b, int* gray_h, int* gray_w);
/* Raw head output of the last forward for parity tests: [anchors][4+nc] fp32 (xywh in network
 * pixels + sigmoid class scores), like the tensor ultralytics' Detect returns. */
int gtx_detector_raw_output(gtx_detector* det, int b, float* out, int* n_anchors);
/* Same layout, class columns hold the pre-sigmoid logits (used to calibrate synthetic weights). */
int gtx_detector_raw_logits(gtx_detector* det, int b, float* out, int* n_anchors);
/* Activation of a named layer of the last forward ("model.4" ...), NHWC fp32, for parity. Refused while a batch is in
 * flight. On the split-f16x3 path with the fused front launch (YOLOv8 n / s) "model.0.conv" and "model.1.conv" are never
 * stored by the forward pass: the call recomputes them with their stand-alone launches (same products, another summation
 * order), so those two dumps are not bit for bit what the network consumed. */
int gtx_detector_layer_output(gtx_detector* det, int b, const char* layer, float* out,
                              int* h, int* w, int* c);
/* fp32_split only. *flag = 1 when an activation of a collected pass (since the last call with clear != 0) lay beyond fp16's
 * range and was clamped to +-65504 where the split-f16x3 path stores it as a (hi, lo) fp16 pair. `half: false` promises
 * fp32's range (default.yaml:245), so gtx_detector_collect / _detect* of THAT pass re-run the batch through an exact-fp32
 * detector built from the same tensors (v_mfma_f32_32x32x2_f32; the frames must still be where the caller put them, which
 * one-batch-in-flight guarantees) and every later pass goes there; gtx_detector_fell_back reports 1 from then on. Trained,
 * BN-folded YOLOv8 weights never get there; GTX_SAT_FALLBACK=0 in the environment keeps the flag and skips the re-run. */
int gtx_detector_saturated(gtx_detector* det, int clear, int* flag);
int gtx_detector_fell_back(gtx_detector* det, int* fell_back);
/* Default fp32 path: the Detect box branch (cv2[l][0], cv2[l][1]) is evaluated at the anchors that pass the score gate only, bit
 * for bit what the dense layers give there (csrc/head_sparse.hip; GTX_SPARSE_BOX=0 builds detectors without it). on = 1 when this
 * detector does so; overflows = collected batches with more candidates per image than its buffer holds (8192), which were
 * finished by the dense layers instead. */
int gtx_detector_sparse_box(gtx_detector* det, int* on, int* overflows);
/* Letterbox-padding rows (ultralytics LetterBox with `rect: false`: 420 + 420 of 1920 input rows for a 16:9 frame): an activation row
 * out of reach of the frame's rows sees the same inputs for every frame, so its value is a constant of the checkpoint. The detector
 * computes those rows once when it is created (one full pass on a blank frame over every batch slot) and its later launches
 * cover the other tile rows only; the buffers keep the constants and every result is the full launches' bit for bit
 * (csrc/detector.cpp plan_pad_skip; GTX_PAD_SKIP=0 builds detectors without it). on = 1 when rows are being skipped; skipped /
 * total = 8-row tile rows left out / launched per image and pass, summed over the convolution launches. */
int gtx_detector_pad_skip(gtx_detector* det, int* on, int* skipped, int* total);
/* gtx_det_config.obj_feats: the appearance vectors of image b of the most recently collected batch, out [n][dim] fp32 in the
 * order of its boxes (n = min(box count, cap); out may be NULL to ask for n and dim). */
int gtx_detector_features(gtx_detector* det, int b, float* out, int cap, int* n, int* dim);

/* Per-kernel-family profile of one forward pass: launches, total ms (HIP events around every
 * launch on the launch stream, graph disabled) and algorithmic FLOPs / bytes. `names` receives
 * up to cap entries of 96 chars. Feeds bench.py's roofline object.
 */
/* Live variant of the profile: after gtx_detector_trace(det, n) every n-th submitted pass carries a HIP event in
 * front of every launch of its forward graph; gtx_detector_profile(det, 0, 0, ...) then returns (and
 * clears) the per-family totals of the traced passes, i.e. kernel durations as they were inside the
 * running pipeline. n = 0 switches tracing off. */
int gtx_detector_trace(gtx_detector* det, int every_n);
int gtx_detector_profile(gtx_detector* det, int nb, int iters, int cap, char* names,
                         int* launches, float* total_ms, double* flops, double* bytes,
                         int* n_families);

/* ------------------------------------------------------------------ tracker (host, C++)
 * Stands in for the tracker callback ultralytics runs inside model.track()
 * (BYTETracker / BOTSORT.update; cfg tracker.* default.yaml:361-389). */

typedef struct gtx_tracker_config {
  int type;                /* 0 = bytetrack, 1 = botsort, 2 = ocsort (default.yaml:391-404), 3 = deepocsort: ocsort + camera-motion
                              compensation by gmc_affine + (with_reid) the appearance term (default.yaml:406-427),
                              4 = fasttrack (default.yaml:426-443): ByteTrack + the occlusion handling of the fields at the end,
                              5 = tracktrack (default.yaml:445-470): multi-cue cost + iterative assignment + track-aware initialisation */
  float track_high_thresh;
  float track_low_thresh;
  float new_track_thresh;
  int track_buffer;
  float match_thresh;
  int fuse_score;
  int frame_rate;          /* ultralytics passes 30 */
  /* OC-SORT only (type 2): tracker.ocsort.{delta_t, inertia, use_byte}; min_hits is OC-SORT's own default (3) */
  int delta_t;
  float inertia;
  int use_byte;
  int min_hits;
  /* FastTracker only (type 4): tracker.fasttrack.* of the reference's config, same names and meaning (default.yaml:436-443) */
  int reset_velocity_offset_occ;
  int reset_pos_offset_occ;
  float enlarge_bbox_occ;
  float dampen_motion_occ;
  int active_occ_to_lost_thresh;
  float occ_cover_thresh;
  int occ_reappear_window;
  float init_iou_suppress;
  /* BoT-SORT only (type 1): tracker.botsort.{with_reid, proximity_thresh, appearance_thresh} (default.yaml:376-378). with_reid = 1:
   * gtx_tracker_update_feats must be used; the cost of a pair whose boxes overlap by at least proximity_thresh becomes
   * min(IoU cost, cosine distance / 2) when the latter is <= 1 - appearance_thresh (BOTSORT.get_dists) */
  int with_reid;
  float proximity_thresh;
  float appearance_thresh;
  /* TrackTrack only (type 5): tracker.tracktrack.* of the reference's config, same names and meaning (default.yaml:453-468);
   * penalty_q is accepted and has nothing to act on (the tracker sees the detector's NMS output only) */
  float lost_match_thr;
  float iou_weight, reid_weight, conf_weight, angle_weight;
  float penalty_p, penalty_q, reduce_step;
  float tai_thr;
  int min_track_len;
  /* Deep OC-SORT (type 3) with with_reid = 1 (default.yaml:420-425; `model: auto`: gtx_detector_features through
   * gtx_tracker_update_feats): proximity_thresh / appearance_thresh gate the appearance term of the first association, and
   * alpha_fixed_emb is the base factor of the track vectors' dynamic-alpha EMA (0 = 0.95). TrackTrack (type 5) with with_reid = 1
   * replaces the second HMIoU term of its cost by the cosine distance (reid_weight, default.yaml:456). */
  float alpha_fixed_emb;
} gtx_tracker_config;

int gtx_tracker_create(const gtx_tracker_config* cfg, gtx_tracker** out);
void gtx_tracker_destroy(gtx_tracker* trk);
int gtx_tracker_reset(gtx_tracker* trk);
/* One frame of detections (xyxy, conf, cls; n entries) -> active tracks. Outputs hold up to
 * cap rows: xyxy (Kalman posterior box), track id, score, class, index of the matched
 * detection (out_det_idx; -1 for a row without one: FastTracker, type 4, keeps an occluded track in the output on its
 * prediction with its last score and class -- a caller that indexes the detections with it must test for -1). gmc_affine: optional 2x3 row-major camera-motion matrix (BoT-SORT GMC), NULL =
 * identity. Mirrors BYTETracker.update + the result rewrite in
 * ultralytics/trackers/track.py:on_predict_postprocess_end. */
int gtx_tracker_update(gtx_tracker* trk, int n, const float* xyxy, const float* conf,
                       const int* cls, const double* gmc_affine, int cap, int* n_out,
                       float* out_xyxy, int* out_id, float* out_score, int* out_cls,
                       int* out_det_idx);

/* gtx_tracker_update with one appearance vector per detection (feats [n][feat_dim] fp32, e.g. gtx_detector_features): BOTrack's
 * normalised current vector and its 0.9-EMA, used by the first association and the unconfirmed one when
 * gtx_tracker_config.with_reid is set (ultralytics/trackers/bot_sort.py). feats == NULL behaves like gtx_tracker_update. */
int gtx_tracker_update_feats(gtx_tracker* trk, int n, const float* xyxy, const float* conf, const int* cls,
                             const double* gmc_affine, const float* feats, int feat_dim, int cap, int* n_out,
                             float* out_xyxy, int* out_id, float* out_score, int* out_cls, int* out_det_idx);

/* The sequential half of a frame-sharded run (rank 0): n_recs per-frame records in clip order, each `stride`
 * doubles laid out as geotrax_amd/distributed.py::pack_frame_record writes them -- n, max_det x (x1, y1, x2, y2, conf, cls),
 * [with_gmc: valid, 2x3 camera-motion warp,] valid, h11..h33 -- go through gtx_tracker_update one after the other
 * (tracker.update on every frame, empty or not; the record's warp for BoT-SORT). rows_per_frame[f] = tracks of frame f;
 * the row_* arrays (row_cap rows) receive them back to back, as gtx_tracker_update would have returned them. */
int gtx_tracker_replay(gtx_tracker* trk, const double* recs, int n_recs, int stride, int max_det, int with_gmc,
                       int row_cap, int* rows_per_frame, float* row_xyxy, int* row_id, float* row_score,
                       int* row_cls, int* row_det_idx);

/* The trackers' assignment solver on its own (host only, no GPU): what `lap.lapjv(cost, extend_cost=True, cost_limit=L)`
 * returns for ultralytics/trackers/utils/matching.py:linear_assignment (lapx, pyproject.toml:58) when cost_limit > 0 -- a pair is
 * matched only below L, an unmatched row or column costs L/2 -- and the plain minimum-cost assignment of min(rows, cols) pairs
 * (scipy.optimize.linear_sum_assignment) when cost_limit <= 0. cost: rows x cols row-major; row_to_col[r] = column or -1;
 * col_to_row (may be NULL) the inverse. Exposed so that tests can hold the solver against scipy on its own. */
int gtx_op_linear_assignment(const float* cost, int rows, int cols, double cost_limit, int* row_to_col, int* col_to_row);

/* ------------------------------------------------------------------ stabilizer
 * Stands in for stabilo.Stabilizer as used at extract.py:139,177-187 and
 * geotrax/utils/registration.py:59-85. */

typedef struct gtx_stab_config {
  float downsample_ratio;      /* stabilo downsample_ratio (default.yaml:106) */
  int max_features;            /* per frame; the reference frame gets ref_multiplier x */
  float ref_multiplier;
  float filter_ratio;          /* Lowe ratio */
  float ransac_threshold;      /* px, full-resolution units */
  int ransac_max_iter;
  float ransac_confidence;
  int mask_use;
  float mask_margin_ratio;
  int fast_threshold;          /* ORB fastThreshold (OpenCV default 20) */
  int n_levels;                /* ORB pyramid levels (8) */
  float scale_factor;          /* ORB pyramid scale (1.2) */
  uint32_t seed;               /* RANSAC sampling seed */
  int frame_h, frame_w;
  int clahe;                   /* stabilo clahe (default.yaml:105): cv2.createCLAHE(2.0, (8, 8)) on the working gray image */
  int affine;                  /* stabilo transformation_type (default.yaml:121): 0 = projective, 1 = affine (3-point samples,
                                  six-parameter refit; the matrix handed back is 3x3 with last row 0 0 1) */
  int filter_type;             /* stabilo filter_type (default.yaml:117): 0 = ratio (Lowe, filter_ratio), 1 = none (every query
                                  keypoint's nearest neighbour goes to the estimator) */
} gtx_stab_config;

int gtx_stabilizer_create(gtx_ctx* ctx, const gtx_stab_config* cfg, gtx_stabilizer** out);
/* The stabilizer's optional pre-processing step on its own: cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(gray)
 * (stabilo `clahe: true`, reference default.yaml:105). gray, out: u8 [h,w] on the host. */
int gtx_op_clahe(gtx_ctx* ctx, const uint8_t* gray, int h, int w, uint8_t* out);
void gtx_stabilizer_destroy(gtx_stabilizer* st);
/* Stabilizer.set_ref_frame(frame, boxes): boxes xywh [n,4] in frame pixels or NULL. */
int gtx_stabilizer_set_ref_frame(gtx_stabilizer* st, const uint8_t* frame_bgr, int h, int w,
                                 const float* boxes_xywh, int n);
/* Same, from a half-resolution gray image already in HBM (gtx_detector_gray). */
int gtx_stabilizer_set_ref_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw,
                                    const float* boxes_xywh, int n);
/* Stabilizer.stabilize(frame, boxes) + get_cur_trans_matrix(): H maps current-frame pixels
 * to reference-frame pixels (row-major 3x3 f64). valid = 0 when no transform could be
 * estimated (the reference then gets None and skips the row, extract.py:185). stats[4] =
 * keypoints ref, keypoints cur, good matches, inliers (registration.py:83-85). */
int gtx_stabilizer_stabilize(gtx_stabilizer* st, const uint8_t* frame_bgr, int h, int w,
                             const float* boxes_xywh, int n, double H[9], int* valid,
                             int stats[4]);
int gtx_stabilizer_stabilize_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw,
                                      const float* boxes_xywh, int n, double H[9], int* valid,
                                      int stats[4]);
/* Asynchronous pair of _stabilize_gray_dev for pipelining: _submit enqueues keypoints, matching and
 * RANSAC on the stabilizer's stream and returns; _collect waits and runs the host refit. */
int gtx_stabilizer_submit_gray_dev(gtx_stabilizer* st, const void* gray_dptr, int gh, int gw,
                                   const float* boxes_xywh, int n);
int gtx_stabilizer_collect(gtx_stabilizer* st, double H[9], int* valid, int stats[4]);
/* The features of the frame stabilized last become the reference (buffers swapped; needs ref_multiplier = 1): frame-to-frame
 * registration (gmc_method orb) without extracting every frame's features twice. */
int gtx_stabilizer_promote_cur(gtx_stabilizer* st);
/* GPU time (ms) of the last collected asynchronous pass: what the reference logs as "Average stabilization
 * time" (geotrax/extract.py:175,188,206) is the wall time of the blocking stabilo calls; here the pass runs on
 * its own stream beside the detector, so its stream-ordered duration is reported instead. */
int gtx_stabilizer_last_ms(gtx_stabilizer* st, float* ms);
/* Keypoints / descriptors of the last processed image (for parity tests): xy in full-res
 * pixels, level, angle bin, 32-byte descriptors. */
int gtx_stabilizer_keypoints(gtx_stabilizer* st, int which /*0 ref, 1 cur*/, int cap, int* n,
                             float* xy, int* level, int* angle_bin, uint8_t* desc);
/* Good matches of the last stabilize call: pairs (cur index, ref index) and Hamming distance. */
int gtx_stabilizer_matches(gtx_stabilizer* st, int cap, int* n, int* cur_idx, int* ref_idx,
                           int* dist);

/* The steered-BRIEF sampling table the descriptor kernel uses: [256 orientation bins][256
 * tests][ax, ay, bx, by] int8 = 262144 bytes. `st` may be NULL (the table does not depend on the
 * object and needs no device). The oracle generates its own table from the published recipe and
 * the parity tests compare the two. */
int gtx_stabilizer_pattern(gtx_stabilizer* st, int8_t* out);

/* ------------------------------------------------------------------ global motion compensation
 * BoT-SORT's camera-motion estimate, gmc_method 'sparseOptFlow' (geotrax/cfg/default.yaml:374), the
 * per-frame step ultralytics' BOTSORT.update runs before association (reached from extract.py:153):
 * Shi-Tomasi corners of the half-resolution gray frame, pyramidal Lucas-Kanade against the previous
 * frame, RANSAC similarity. A = row-major 2x3 f64 mapping previous-frame to current-frame pixels
 * (full resolution); feed it to gtx_tracker_update(gmc). Identity (valid = 0) on the first frame or
 * when fewer than 5 corners could be tracked. stats = {corners of the previous frame, tracked, inliers}. */
typedef struct gtx_gmc gtx_gmc;
int gtx_gmc_create(gtx_ctx* ctx, int frame_h, int frame_w, int seed, gtx_gmc** out);
void gtx_gmc_destroy(gtx_gmc* g);
int gtx_gmc_reset(gtx_gmc* g);
int gtx_gmc_apply(gtx_gmc* g, const uint8_t* frame_bgr, int h, int w, double A[6], int* valid, int stats[3]);
/* Asynchronous pair on the half-resolution gray image the detector left in HBM (gtx_detector_gray).
 * Up to 64 frames may be submitted ahead (the image is copied at submit); _collect returns their warps
 * in submission order. */
int gtx_gmc_submit_gray_dev(gtx_gmc* g, const void* gray_dptr, int gray_h, int gray_w);
/* The next submitted frame opens a new sequence (identity warp); unlike _reset, frames may still be in flight. */
int gtx_gmc_restart(gtx_gmc* g);
/* The same for a full BGR u8 frame [h][w][3] in HBM (gray + 2x2 mean first). restart != 0: the frame opens a new
 * sequence -- its own warp is the identity and the next submitted frame is compensated against it. A rank of the
 * frame-sharded run hands the GMC the frame that precedes its batch in the clip this way. */
int gtx_gmc_submit_frame_dev(gtx_gmc* g, const void* frame_bgr_dptr, int h, int w, int restart);
int gtx_gmc_collect(gtx_gmc* g, double A[6], int* valid, int stats[3]);
/* Parity hook: which 0 = corners of the last frame, 1 = corners of the frame before, 2 = where LK put
 * those in the last frame (+ status). xy in half-resolution pixels. */
int gtx_gmc_points(gtx_gmc* g, int which, int cap, int* n, float* xy, int* status);

/* ------------------------------------------------------------------ camera-motion compensation, method 'ecc'
 * `gmc_method: ecc` (geotrax/cfg/default.yaml:374,419,467): ultralytics' GMC.apply_ecc, i.e. cv2.findTransformECC(first frame,
 * current frame, MOTION_EUCLIDEAN, (EPS | COUNT, max_iters = 5000, eps = 1e-6), None, 1) on cvtColor(BGR2GRAY) -> GaussianBlur(3x3, 1.5)
 * -> resize(1/2). As upstream, every frame is registered against the FIRST frame since the last reset, and A -- row-major 2x3,
 * float32 values -- is in half-resolution pixels (upstream does not scale the translation back for this method). Identity for
 * the first frame. info = {iterations run, status: 0 finished, 1 NaN correlation, 2 the correlation was about to be minimised
 * (both raise cv2.error upstream, which the caller there catches and keeps the matrix as the failed call left it: so does A)}.
 * _submit_dev prepares the frame's image at once on `producer`'s stream (the context whose stream wrote the frame, e.g. the
 * detector's; NULL = the object's own) into a 32-deep ring; _collect returns the frames in submission order. */
typedef struct gtx_ecc gtx_ecc;
int gtx_ecc_create(gtx_ctx* ctx, int frame_h, int frame_w, int max_iters, double eps, gtx_ecc** out);
void gtx_ecc_destroy(gtx_ecc* e);
int gtx_ecc_reset(gtx_ecc* e);
/* replace != 0: every collected frame becomes the template of the next one (frame-to-frame warps). Default 0 = upstream's behaviour. */
int gtx_ecc_replace_template(gtx_ecc* e, int replace);
/* exact != 0 (default): warpAffine's bilinear samples as OpenCV >= 4.11 takes them (source position in floating point); 0: as through
 * 4.10 (fixed point, rounded to 1/32 pixel; many fits then never meet eps and run to max_iters -- oracle/ecc_ref.py). */
int gtx_ecc_exact_positions(gtx_ecc* e, int exact);
int gtx_ecc_submit(gtx_ecc* e, const uint8_t* frame_bgr, int h, int w);
int gtx_ecc_submit_dev(gtx_ecc* e, gtx_ctx* producer, const void* frame_bgr_dptr, int h, int w);
int gtx_ecc_collect(gtx_ecc* e, double A[6], int info[2], double* rho);
/* Parity hook: which 0 = the prepared image of the frame collected last, 1 = the template; out [h / 2][w / 2] float32 */
int gtx_ecc_image(gtx_ecc* e, int which, float* out);

/* ------------------------------------------------------------------ registration (once per video)
 * Replaces estimate_homography() of geotrax/utils/registration.py:21-95 -- stabilo.Stabilizer with
 * detector_name='rsift', matcher_name='bf', filter_type='ratio', projective model, no mask, no
 * downsampling -- as used by the georeference stage for frame <-> master frame <-> orthophoto
 * (SURVEY.md K11). RootSIFT keypoints/descriptors on the GPU, brute-force L2 2-NN on MFMA, Lowe
 * ratio, robust homography (MSAC hypotheses + IRLS refit; the reference uses MAGSAC++). */
typedef struct gtx_reg_config {
  int max_features;         /* SIFT nfeatures: strongest responses kept (reference default 250000) */
  float filter_ratio;       /* Lowe ratio (reference default 0.55) */
  float ransac_threshold;   /* reprojection threshold in destination pixels (3.0) */
  int ransac_max_iter;      /* hypotheses (10000; clamped to [256, 16384]) */
  float ransac_confidence;  /* accepted for interface parity; the hypothesis count is fixed */
  float rsift_eps;          /* RootSIFT L1-normalisation epsilon (1e-8); negative: plain SIFT descriptors (stabilo `detector_name: sift`) */
  int seed;
} gtx_reg_config;
/* src/dst: BGR u8 [h][w][3] host images. H (row-major 3x3 f64) maps src pixels to dst pixels;
 * stats = {n_src_keypoints, n_dst_keypoints, n_good_matches, n_inliers}; *valid = 0 when no model
 * was found (the reference then retries with half the features, registration.py:87-91);
 * timings_ms (may be NULL) = {detect+describe both images, matching, ratio filter, robust fit}. */
int gtx_register_images(gtx_ctx* ctx, const gtx_reg_config* cfg, const uint8_t* src_bgr, int src_h,
                        int src_w, const uint8_t* dst_bgr, int dst_h, int dst_w, double H[9],
                        int* valid, int stats[4], float timings_ms[4]);
/* The detector stage on its own (cv2.SIFT_create(nfeatures, enable_precise_upscale=True)
 * .detectAndCompute + stabilo's RootSIFT conversion when root != 0), for parity tests: keypoints as
 * rows {x, y, size, angle, response} + the packed octave word, descriptors [n][128] fp32. */
typedef struct gtx_sift gtx_sift;
int gtx_sift_create(gtx_ctx* ctx, int max_h, int max_w, gtx_sift** out);
void gtx_sift_destroy(gtx_sift* s);
int gtx_sift_detect(gtx_sift* s, const uint8_t* image_bgr, int h, int w, int max_features, int root,
                    float root_eps, int cap, int* n, float* kp5, int* octave, float* desc);
/* GPU milliseconds of the stages of the last detect call (HIP events on the context's stream): out[0] upload + gray + Gaussian / DoG
 * pyramid, out[1] extrema + refinement + orientation (with the host round trips for their counters), out[2] descriptors; out[3] = the
 * pixel count of the doubled base image (the pyramid is 11 x 4/3 fp32 images of that size: what a roofline of the stage is priced on). */
int gtx_sift_stage_ms(gtx_sift* s, float out[4]);
/* Gaussian (kind 0) or DoG (kind 1) image of the last detect call, [h][w] fp32. */
int gtx_sift_pyramid(gtx_sift* s, int kind, int octave, int layer, int cap, float* out, int* h, int* w,
                     int* n_octaves);

/* Stabilizer.transform_cur_boxes(): maps the 4 corners of each xywh box through H and
 * returns the axis-aligned bounding rectangle as xywh (rule pinned on the reference's golden
 * output). Pure host arithmetic, f64 inside, f32 out. */
int gtx_warp_boxes(const double H[9], const float* xywh_in, int n, float* xywh_out);

/* cv2.perspectiveTransform on N points (georeference.py:599-605), f64. */
int gtx_perspective_points(const double H[9], const double* x, const double* y, int n,
                           double* ox, double* oy);

/* cv2.estimateAffinePartial2D(prev, cur, RANSAC) as ultralytics' GMC calls it for `gmc_method: orb` / `sift` (default.yaml:374):
 * the 4-parameter similarity p -> q of n matched points ([n][2] float32 each), A row-major 2x3 f64; *valid = 0 when no model
 * exists (fewer than two distinct points). Host code (a few hundred matches); no device needed. */
int gtx_op_estimate_affine_partial(const float* p_xy, const float* q_xy, int n, unsigned seed, double A[6], int* valid, int* n_inliers);

/* The georeference stage's per-row transform chain in one HIP pass (replaces
 * geotrax/georeference.py:173-177 = apply_homography :599-605 -> ortho2geo :608-615 -> geo2local :618-628, where the
 * reference reprojects through pyproj): frame pixel -H-> orthophoto pixel -affine-> lat/lon (deg)
 * -transverse Mercator (Krueger series, 6th order)-> metres. f64. x, y and every non-NULL output are host arrays of n. */
typedef struct gtx_georef_chain {
  double H[9];            /* frame (stabilized) pixel -> orthophoto pixel */
  double ortho[6];        /* lng0, lat0, dlng, dlat, skew_x, skew_y (georeference.py:608-615) */
  int projected;          /* 0: stop at lat/lon; east/north are not written */
  double semi_major, flattening;            /* ellipsoid of the target CRS */
  double lon0_deg, k0;                      /* central meridian, scale on it */
  double false_easting, false_northing;     /* false_northing includes -k0 * (meridian arc to the latitude of origin) */
} gtx_georef_chain;
int gtx_op_georef_points(gtx_ctx* ctx, const gtx_georef_chain* chain, const double* x, const double* y, int n,
                         double* ortho_x, double* ortho_y, double* lat, double* lon, double* east, double* north);

/* cv2.warpPerspective(frame, H, (w,h)) with bilinear sampling and constant-0 border
 * (visualize.py:289). BGR u8 in/out, host buffers. */
int gtx_warp_frame(gtx_ctx* ctx, const uint8_t* src_bgr, int h, int w, const double H[9],
                   uint8_t* dst_bgr);
/* Same, both images resident in HBM (dptrs from gtx_dev_alloc, distinct buffers); enqueued on the
 * context's stream, returns without waiting (gtx_ctx_synchronize / a later call on the stream orders it). */
int gtx_warp_frame_dev(gtx_ctx* ctx, const void* src_dptr, int h, int w, const double H[9], void* dst_dptr);

/* ------------------------------------------------------------------ result files (host code, no GPU work)
 *
 * The text tables the two stages end with, byte for byte as the reference writes them, formatted on a few threads.
 * gtx_write_table_f32 / _f64 replace np.savetxt(path, table, fmt='%.<precision>g', delimiter=',') of save_results
 * (geotrax/extract.py:497-516: the tracks with '%g' = precision 6 on float32 rows, the transforms with '%.16g' on float64 rows;
 * save_homography's '%.20g' line, georeference.py:879-889). data: rows x cols, row-major. n_threads <= 0: up to 8.
 * gtx_write_csv replaces pandas.DataFrame.to_csv(path, index=False) for the georeferenced table (georeference.py:802-877):
 * header_line = the column names joined by commas; kinds[c]: 0 = int64 column, 1 = float64 column written as repr(float)
 * (shortest round-trip digits, NaN = empty cell), 2 = int32 codes into categories[c][0 .. n_categories[c]) (strings already
 * quoted as a CSV cell needs; a negative code = missing = empty cell; categories / n_categories may be NULL without such columns).
 * columns[c]: `rows` values of the column's kind. */
int gtx_write_table_f32(const char* path, const float* data, int64_t rows, int cols, int precision, int n_threads);
int gtx_write_table_f64(const char* path, const double* data, int64_t rows, int cols, int precision, int n_threads);
int gtx_write_csv(const char* path, const char* header_line, int n_cols, const int* kinds, const void* const* columns,
                  const char* const* const* categories, const int* n_categories, int64_t rows, int n_threads);
/* The per-row walk of estimate_vehicle_dimensions (geotrax/extract.py:433-452) for every track of a table in one call (host code).
 * xc, yc: the observations' centres, float32, the rows of track t at [start[t], start[t+1]). is_anchor[row] = 1 where the reference's
 * loop finds the observation at least `radius` pixels from the current anchor (and makes it the next one); step_dx / step_dy[row]:
 * that step, for the caller's azimuth test. The same float32 operation sequence as the NumPy scalars of the reference (csrc/table_writer.cpp). */
int gtx_track_anchor_walk(const float* xc, const float* yc, const int64_t* start, int n_tracks, float radius,
                          uint8_t* is_anchor, float* step_dx, float* step_dy);

#ifdef __cplusplus
}
#endif
#endif /* GTX_H_ */
