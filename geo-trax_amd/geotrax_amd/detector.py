"""Detector object over the C ABI (gtx_detector_*): the predict half of ``model.track()``.

Reference behaviour replaced: ultralytics' predictor pipeline reached from
geotrax/extract.py:153 -- LetterBox + normalise, YOLOv8 forward, confidence filter,
class-agnostic NMS, scaling back to frame pixels -- with the config keys the reference passes
(geotrax/cfg/default.yaml:229-262: imgsz, conf, iou, max_det, classes, agnostic_nms, half, rect).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import DetConfig, check, ptr

logger = logging.getLogger(__name__)


# which fp32-grade convolution a half=False detector uses unless told otherwise (see Detector.__init__)
FP32_SPLIT_DEFAULT = True


@dataclass
class Detections:
    xyxy: np.ndarray   # [n,4] float32, frame pixels
    conf: np.ndarray   # [n] float32
    cls: np.ndarray    # [n] int32
    speed: dict        # {'preprocess','inference','postprocess'} ms, like results[0].speed
    feats: np.ndarray | None = None   # [n, dim] float32 appearance vectors (Detector(obj_feats=True)), what BoT-SORT's `model: auto` ReID reads

    def __len__(self):
        return len(self.conf)

    @property
    def xywh(self) -> np.ndarray:
        b = self.xyxy
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32) \
            if len(b) else np.zeros((0, 4), np.float32)


class Detector:
    def __init__(self, tensors: dict[str, np.ndarray], frame_hw: tuple[int, int], *, imgsz: int = 1920,
                 conf: float = 0.25, iou: float = 0.7, max_det: int = 1000, classes=None,
                 agnostic_nms: bool = True, half: bool = False, rect: bool = False, max_batch: int = 1,
                 fp32_split: bool | None = None, obj_feats: bool = False, ctx: _lib.Context | None = None):
        """obj_feats: keep one appearance vector per box (Detections.feats; include/gtx.h gtx_det_config.obj_feats).
        half=False (the reference default, default.yaml:245) computes at fp32 grade: fp32 activations in HBM and
        either the exact-fp32 MFMA (fp32_split=False) or the split-f16x3 convolutions (fp32_split=True: hi + lo fp16
        operands, three fp16 MFMAs per product, fp32 accumulate; csrc/conv_igemm_split.hip). fp32_split=None takes
        GTX_FP32_SPLIT from the environment (default: FP32_SPLIT_DEFAULT)."""
        if fp32_split is None:
            fp32_split = os.environ.get("GTX_FP32_SPLIT", "1" if FP32_SPLIT_DEFAULT else "0") == "1"
        self.fp32_split = bool(fp32_split) and not half
        self.ctx = ctx or _lib.default_context()
        lib = self.ctx.lib
        from .weights import is_rtdetr

        self.rtdetr = is_rtdetr(tensors)     # the graph the tensors describe picks the detector family (reference: the model's yaml, extract.py:222-225)
        if self.rtdetr and obj_feats:
            raise NotImplementedError("RT-DETR: obj_feats (ReID `model: auto`) is not implemented")
        nc = int(tensors["model.28.enc_score_head.weight" if self.rtdetr else "model.22.cv3.0.2.weight"].shape[0])
        cfg = DetConfig(imgsz=imgsz, conf=conf, iou=iou, max_det=max_det, agnostic_nms=int(agnostic_nms),
                        half=int(half), rect=int(rect), nc=nc, n_classes=0, max_batch=max_batch,
                        frame_h=frame_hw[0], frame_w=frame_hw[1], fp32_split=int(self.fp32_split), obj_feats=int(bool(obj_feats)), arch=int(self.rtdetr))
        self.obj_feats = bool(obj_feats)
        if classes is not None:
            classes = list(classes)
            cfg.n_classes = len(classes)
            for i, c in enumerate(classes):
                cfg.classes[i] = int(c)
        self.max_det, self.max_batch, self.nc, self.frame_hw = max_det, max_batch, nc, tuple(frame_hw)
        h = C.c_void_p()
        check(lib.gtx_detector_create(self.ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle = h
        for name, arr in tensors.items():
            if name.endswith("dfl.conv.weight") or ".bn." in name:
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(lib.gtx_detector_set_tensor(h, name.encode(), ptr(a), a.ndim, shape))
        check(lib.gtx_detector_finalize(h))
        nh, nw = C.c_int(), C.c_int()
        check(lib.gtx_detector_input_size(h, C.byref(nh), C.byref(nw)))
        self.net_hw = (nh.value, nw.value)
        self._n = np.zeros(max_batch, np.int32)
        self._xyxy = np.zeros((max_batch, max_det, 4), np.float32)
        self._conf = np.zeros((max_batch, max_det), np.float32)
        self._cls = np.zeros((max_batch, max_det), np.int32)
        self._speed = np.zeros(3, np.float32)
        self._sat_warned = False
        if self.rtdetr:
            meta = tensors.get("rtdetr.meta")
            self.n_queries = int(meta[2]) if meta is not None else 300

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.gtx_detector_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def saturated(self, clear: bool = False) -> bool:
        """fp32_split only: an activation of a collected pass lay beyond fp16's range where the split-f16x3 path stores it
        (include/gtx.h: gtx_detector_saturated). The library then re-ran that batch through the exact-fp32 kernels and stays
        there (fell_back())."""
        if not self.fp32_split:
            return False
        f = C.c_int()
        check(self.ctx.lib.gtx_detector_saturated(self.handle, int(clear), C.byref(f)))
        return bool(f.value)

    def pad_skip(self) -> tuple[bool, int, int]:
        """(on, skipped, total): whether this detector leaves the frame-independent rows of the letterbox padding out of its
        launches (computed once at creation; GTX_PAD_SKIP=0: off), and how many 8-row tile rows per image and pass that is, of how
        many (include/gtx.h: gtx_detector_pad_skip)."""
        on, sk, tot = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_detector_pad_skip(self.handle, C.byref(on), C.byref(sk), C.byref(tot)))
        return bool(on.value), int(sk.value), int(tot.value)

    def sparse_box(self) -> tuple[bool, int]:
        """(on, overflows): whether this detector evaluates the Detect box branch at the candidate anchors only (the default fp32
        path; GTX_SPARSE_BOX=0 at construction: off) and how many collected batches had more candidates than its buffer holds and
        were finished by the dense layers (include/gtx.h: gtx_detector_sparse_box)."""
        on, over = C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_detector_sparse_box(self.handle, C.byref(on), C.byref(over)))
        return bool(on.value), int(over.value)

    def fell_back(self) -> bool:
        """True once a saturating split-f16x3 pass has moved this detector to the exact-fp32 convolutions."""
        f = C.c_int()
        check(self.ctx.lib.gtx_detector_fell_back(self.handle, C.byref(f)))
        return bool(f.value)

    def _collect(self, nb: int) -> list[Detections]:
        if self.fp32_split and not self._sat_warned and self.saturated():
            self._sat_warned = True
            if self.fell_back():
                logger.warning("activations beyond fp16's range (|x| > 65504) reached the split-f16x3 convolutions: the batch was re-run with the "
                               "exact-fp32 MFMA convolutions and the detector stays on them for the rest of the run (same results as fp32_split: "
                               "false, at its speed); set fp32_split: false for this checkpoint to skip the detour")
            else:
                logger.warning("activations beyond fp16's range (|x| > 65504) were clamped by the split-f16x3 convolutions and GTX_SAT_FALLBACK=0 "
                               "keeps them: detections differ from an fp32 run of this checkpoint")
        sp = dict(preprocess=float(self._speed[0]), inference=float(self._speed[1]), postprocess=float(self._speed[2]))
        out = []
        for b in range(nb):
            n = int(self._n[b])
            out.append(Detections(self._xyxy[b, :n].copy(), self._conf[b, :n].copy(), self._cls[b, :n].copy(), sp,
                                  self.features(b, n) if self.obj_feats else None))
        return out

    def features(self, b: int = 0, n: int | None = None) -> np.ndarray:
        """[n, dim] appearance vectors of image b of the batch collected last (gtx_detector_features)."""
        cnt, dim = C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_detector_features(self.handle, b, None, self.max_det, C.byref(cnt), C.byref(dim)))
        n = cnt.value if n is None else min(n, cnt.value)
        out = np.zeros((n, dim.value), np.float32)
        if n:
            check(self.ctx.lib.gtx_detector_features(self.handle, b, ptr(out), n, C.byref(cnt), C.byref(dim)))
        return out

    def detect(self, frame_bgr: np.ndarray) -> Detections:
        """One host frame (HxWx3 BGR uint8)."""
        f = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
        h, w, _ = f.shape
        check(self.ctx.lib.gtx_detector_detect(self.handle, ptr(f), h, w, ptr(self._n), ptr(self._xyxy),
                                               ptr(self._conf), ptr(self._cls), ptr(self._speed)))
        return self._collect(1)[0]

    def detect_dev(self, frames_dptr: int, nb: int = 1) -> list[Detections]:
        """nb frames already resident in HBM, back to back."""
        h, w = self.frame_hw
        check(self.ctx.lib.gtx_detector_detect_batch_dev(self.handle, C.c_void_p(frames_dptr), nb, h, w, ptr(self._n),
                                                         ptr(self._xyxy), ptr(self._conf), ptr(self._cls), ptr(self._speed)))
        return self._collect(nb)

    def submit_dev(self, frames_dptr: int, nb: int = 1) -> None:
        """Enqueue a batch (frames resident in HBM) and return at once; pair with collect()."""
        h, w = self.frame_hw
        check(self.ctx.lib.gtx_detector_submit_dev(self.handle, C.c_void_p(frames_dptr), nb, h, w))
        self._flight = nb

    def collect(self) -> list[Detections]:
        check(self.ctx.lib.gtx_detector_collect(self.handle, ptr(self._n), ptr(self._xyxy), ptr(self._conf), ptr(self._cls),
                                                ptr(self._speed)))
        return self._collect(self._flight)

    def gray_dptr(self, b: int = 0) -> tuple[int, int, int]:
        gh, gw = C.c_int(), C.c_int()
        p = self.ctx.lib.gtx_detector_gray(self.handle, b, C.byref(gh), C.byref(gw))
        return p, gh.value, gw.value

    def raw_output(self, b: int = 0, logits: bool = False) -> np.ndarray:
        """[anchors, 4+nc] fp32 of the last forward (xywh network pixels + class scores, or the
        pre-sigmoid class logits)."""
        na = C.c_int()
        h, w = self.net_hw
        anchors = (h // 8) * (w // 8) + (h // 16) * (w // 16) + (h // 32) * (w // 32)
        if self.rtdetr:                      # [queries, 4 + nc]: xywh normalised to the frame + class scores (or logits)
            anchors = self.n_queries
        out = np.zeros((anchors, 4 + self.nc), np.float32)
        fn = self.ctx.lib.gtx_detector_raw_logits if logits else self.ctx.lib.gtx_detector_raw_output
        check(fn(self.handle, b, ptr(out), C.byref(na)))
        assert na.value == anchors
        return out

    def layer_output_int(self, layer: str, b: int = 0) -> np.ndarray:
        """An integer read-back (RT-DETR's selected anchor indices, layer 'model.28.topk')."""
        return self.layer_output(layer, b).view(np.int32)

    def layer_output(self, layer: str, b: int = 0) -> np.ndarray:
        h, w, c = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_detector_layer_output(self.handle, b, layer.encode(), None, C.byref(h), C.byref(w), C.byref(c)))
        out = np.zeros((h.value, w.value, c.value), np.float32)
        check(self.ctx.lib.gtx_detector_layer_output(self.handle, b, layer.encode(), ptr(out), C.byref(h), C.byref(w), C.byref(c)))
        return out

    def trace(self, every_n: int) -> None:
        """Time every launch of every `every_n`-th submitted pass with HIP events (0 = off)."""
        check(self.ctx.lib.gtx_detector_trace(self.handle, every_n))

    def trace_report(self) -> list[dict]:
        """Per-kernel-family totals of the traced passes since the last report."""
        return self.profile(nb=0, iters=0)

    def profile(self, nb: int = 1, iters: int = 5) -> list[dict]:
        """Per-kernel-family totals of `iters` forward passes (HIP events around every launch)."""
        cap = 64
        names = C.create_string_buffer(cap * 96)
        launches = np.zeros(cap, np.int32)
        ms = np.zeros(cap, np.float32)
        flops = np.zeros(cap, np.float64)
        nbytes = np.zeros(cap, np.float64)
        n = C.c_int()
        check(self.ctx.lib.gtx_detector_profile(self.handle, nb, iters, cap, names, ptr(launches), ptr(ms), ptr(flops),
                                                ptr(nbytes), C.byref(n)))
        out = []
        for i in range(n.value):
            nm = names.raw[i * 96:(i + 1) * 96].split(b"\0")[0].decode()
            out.append(dict(kernel=nm, launches=int(launches[i]), total_ms=float(ms[i]), flops=float(flops[i]),
                            bytes=float(nbytes[i])))
        return out
