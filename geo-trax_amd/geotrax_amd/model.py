"""``YOLO``: the detector+tracker object the extract loop calls, with ultralytics' call contract.

Reference usage replaced (geotrax/extract.py:153-168, 222):

    model = YOLO(model=path, task='detect')
    results = model.track(frame, **config['ultralytics'], persist=True)
    boxes = results[0].boxes          # .id (or None) / .xywh / .cls / .conf, len(boxes)
    speed = results[0].speed          # {'preprocess','inference','postprocess'} in ms

``track`` runs the HIP detector (geotrax_amd.detector.Detector) and the host C++ tracker
(geotrax_amd.tracker.Tracker) and rewrites the boxes with the tracker output exactly like
ultralytics' ``on_predict_postprocess_end`` callback: rows become the Kalman-posterior boxes of the
active tracks, ``id`` is set, detections the tracker has not confirmed disappear; when the tracker
returns nothing the raw detections are kept with ``id = None`` (the reference then writes -1 and
drops the rows, extract.py:161-165, 287).

Arrays come back as numpy; they also answer ``.detach()``, ``.cpu()`` and ``.numpy(force=True)`` so
the unmodified reference loop can consume them.
"""
from __future__ import annotations

import logging
from pathlib import Path

import numpy as np
import yaml

from . import _lib
from .detector import Detector
from .tracker import TRACKER_TYPES, Tracker
from .weights import load_weights

logger = logging.getLogger(__name__)


class _Arr(np.ndarray):
    """ndarray that tolerates the torch idioms of the reference loop."""

    def detach(self):
        return self

    def cpu(self):
        return self

    def numpy(self, force: bool = False):
        return np.asarray(self)


def _arr(a) -> _Arr:
    return np.asarray(a).view(_Arr)


class Boxes:
    def __init__(self, xyxy: np.ndarray, conf: np.ndarray, cls: np.ndarray, ids: np.ndarray | None):
        self._xyxy = np.asarray(xyxy, dtype=np.float32).reshape(-1, 4)
        self._conf = np.asarray(conf, dtype=np.float32)
        self._cls = np.asarray(cls, dtype=np.float32)
        self._id = None if ids is None else np.asarray(ids, dtype=np.float32)

    def __len__(self):
        return len(self._conf)

    @property
    def xyxy(self):
        return _arr(self._xyxy)

    @property
    def xywh(self):
        b = self._xyxy
        return _arr(np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1)
                    .astype(np.float32) if len(b) else np.zeros((0, 4), np.float32))

    @property
    def conf(self):
        return _arr(self._conf)

    @property
    def cls(self):
        return _arr(self._cls)

    @property
    def id(self):
        return None if self._id is None else _arr(self._id)

    @property
    def is_track(self):
        return self._id is not None


class Results:
    def __init__(self, boxes: Boxes, speed: dict, orig_shape: tuple, names: dict):
        self.boxes, self.speed, self.orig_shape, self.names = boxes, speed, orig_shape, names

    def __len__(self):
        return len(self.boxes)


_PREDICT_KEYS = ("imgsz", "conf", "iou", "max_det", "classes", "agnostic_nms", "half", "rect")


class YOLO:
    def __init__(self, model, task: str = "detect", ctx: _lib.Context | None = None):
        """model: path to a ``.safetensors`` weight file (see tools/convert_weights.py) or a dict of
        tensors. An ultralytics ``.pt`` pickle cannot be read without ultralytics."""
        if task not in (None, "detect"):
            raise NotImplementedError(f"task='{task}': only 'detect' is implemented")
        self.ctx = ctx
        self.names: dict[int, str] = {}
        if isinstance(model, dict):
            self.tensors = model
            self.model_path = None
        else:
            p = Path(model)
            if p.suffix == ".pt":
                raise ValueError(f"'{p}' is an ultralytics pickle; convert it once with tools/convert_weights.py "
                                 "(needs ultralytics) and point cfg -> extraction -> model at the .safetensors file")
            self.tensors = load_weights(p)
            self.model_path = p
            side = p.with_suffix(".names.yaml")
            if side.is_file():
                self.names = {int(k): str(v) for k, v in yaml.safe_load(side.read_text()).items()}
        from .weights import is_rtdetr

        self.is_rtdetr = is_rtdetr(self.tensors)
        nc = int(self.tensors["model.28.enc_score_head.weight" if self.is_rtdetr else "model.22.cv3.0.2.weight"].shape[0])
        if not self.names:
            self.names = {i: str(i) for i in range(nc)}
        self.model = self            # ultralytics exposes .model.yaml_file; keep attribute access harmless
        # the reference reads this attribute to swap YOLO for RTDETR (extract.py:223-225); here one class serves both graphs
        self.yaml_file = "rtdetr-l.yaml" if self.is_rtdetr else "yolov8.yaml"
        self._det: Detector | None = None
        self._det_key = None
        self._tracker: Tracker | None = None
        self._tracker_key = None
        self._gmc = None
        self._gmc_method = None
        self.fp32_split: bool | None = None    # half: false only; None = the library default (split-f16x3), see Detector

    # ---- lazy construction, like ultralytics' predictor setup on the first call
    def _detector(self, frame_hw, kw) -> Detector:
        if kw.get("augment"):          # ultralytics.augment (default.yaml:243): test-time augmentation changes the detections
            raise NotImplementedError("augment=True (test-time augmentation) is not implemented")
        key = (tuple(frame_hw), self.fp32_split, bool(getattr(self, "_obj_feats", False))) + tuple(repr(kw.get(k)) for k in _PREDICT_KEYS)
        if self._det is None or key != self._det_key:
            if self._det is not None:
                self._det.close()
            imgsz = kw.get("imgsz", 640)
            if isinstance(imgsz, (list, tuple)):
                imgsz = max(imgsz)
            self._det = Detector(self.tensors, frame_hw, imgsz=int(imgsz), conf=float(kw.get("conf") or 0.1),
                                 iou=float(kw.get("iou", 0.7)), max_det=int(kw.get("max_det", 300)), classes=kw.get("classes"),
                                 agnostic_nms=bool(kw.get("agnostic_nms", False)), half=bool(kw.get("half", False)),
                                 rect=bool(kw.get("rect", False)), fp32_split=self.fp32_split, obj_feats=bool(getattr(self, "_obj_feats", False)), ctx=self.ctx)   # absent -> the reference config's value (default.yaml:300)
            self._det_key = key
        return self._det

    def _make_tracker(self, spec) -> Tracker:
        if isinstance(spec, (str, Path)):
            params = yaml.safe_load(Path(spec).read_text())
        else:
            params = dict(spec or {})
        ttype = params.get("tracker_type", "botsort")
        if ttype not in TRACKER_TYPES:
            raise NotImplementedError(f"tracker_type '{ttype}' is not implemented (available: {sorted(TRACKER_TYPES)})")
        if ttype in ("botsort", "deepocsort", "tracktrack"):  # the trackers that take a camera-motion warp per frame
            if params.get("with_reid"):
                # `model: auto` (default.yaml:379, :421, :470): appearance vectors from the detector's own feature maps
                # (Detector(obj_feats=True) -> Tracker.update(feats=)); a separate ReID network's weights cannot be read here
                if str(params.get("model", "auto")) != "auto":
                    raise NotImplementedError(f"{ttype} with_reid: only `model: auto` (detector-derived features) is implemented, not '{params.get('model')}'")
            gm = params.get("gmc_method", "none")
            if gm in ("none", None):
                self._gmc_method = None
            elif gm in ("sparseOptFlow", "orb", "sift", "ecc"):
                self._gmc_method = gm                         # sparseOptFlow: GPU corners + pyramidal LK + RANSAC similarity; orb / sift: gmc.FeatureGMC; ecc: gmc.EccGMC
            else:
                raise NotImplementedError(f"{ttype} gmc_method '{gm}': the reference's choices are 'sparseOptFlow', 'orb', 'sift', 'ecc' and 'none'")
        else:
            self._gmc_method = None
        self._gmc = None
        if ttype == "tracktrack" and not getattr(self, "_tracktrack_warned", False):
            self._tracktrack_warned = True
            logger.warning("tracker 'tracktrack': written from the description of its parameters in the config and the published method (height-modulated "
                           "IoU + confidence + corner-angle cost, iterative mutual-minimum assignment under a shrinking threshold, track-aware "
                           "initialisation); with_reid: true uses the detector's own vectors (`model: auto`; otherwise reid_weight falls back to the HMIoU distance), penalty_q has nothing to act on; the "
                           "pinned ultralytics' implementation may differ where that description leaves a choice open (oracle/tracktrack_ref.py lists "
                           "the choices). Score a reference run with tools/score_run.py to pin it.")
        if ttype == "fasttrack" and not getattr(self, "_fasttrack_warned", False):
            self._fasttrack_warned = True
            logger.warning("tracker 'fasttrack': written from the description of its parameters in the config (occlusion test by box cover, "
                           "Kalman roll-back at the onset, dampened velocity, one-shot box enlargement, re-find window, init-IoU suppression on "
                           "top of ByteTrack); the pinned ultralytics' implementation may differ where that description leaves a choice open "
                           "(oracle/fasttrack_ref.py lists the choices). Score a reference run with tools/score_run.py to pin it.")
        if ttype in ("ocsort", "deepocsort") and not getattr(self, "_ocsort_warned", False):
            self._ocsort_warned = True
            logger.warning(f"tracker '{ttype}': this build follows the authors' published OC-SORT (observation-centric re-update, momentum, "
                           "recovery, optional BYTE pass); how the pinned ultralytics maps its config keys onto it is an assumption "
                           "(det_thresh = track_high_thresh, iou_threshold = 1 - match_thresh, max_age = track_buffer, min_hits = 3 unless "
                           f"'min_hits' is given) and 'fuse_score'{' (set in this config)' if params.get('fuse_score') else ''} is ignored: "
                           "tracks may differ from a geo-trax run of the same tracker. Score a reference run with tools/score_run.py to pin it.")
        return Tracker(ttype, **{k: v for k, v in params.items() if k in (
            "track_high_thresh", "track_low_thresh", "new_track_thresh", "track_buffer", "match_thresh", "fuse_score",
            "delta_t", "inertia", "use_byte", "min_hits", "reset_velocity_offset_occ", "reset_pos_offset_occ", "enlarge_bbox_occ",
            "dampen_motion_occ", "active_occ_to_lost_thresh", "occ_cover_thresh", "occ_reappear_window", "init_iou_suppress",
            "with_reid", "proximity_thresh", "appearance_thresh", "lost_match_thr", "iou_weight", "reid_weight", "conf_weight", "angle_weight",
            "penalty_p", "penalty_q", "reduce_step", "tai_thr", "min_track_len", "alpha_fixed_emb")})

    # ---- ultralytics-style entry points
    def predict(self, source: np.ndarray, **kwargs) -> list[Results]:
        frame = np.ascontiguousarray(source, dtype=np.uint8)
        det = self._detector(frame.shape[:2], kwargs)
        d = det.detect(frame)
        return [Results(Boxes(d.xyxy, d.conf, d.cls, None), d.speed, frame.shape[:2], self.names)]

    def track(self, source: np.ndarray, persist: bool = False, **kwargs) -> list[Results]:
        spec = kwargs.get("tracker", {"tracker_type": "botsort"})
        tkey = repr(spec)
        if self._tracker is None or tkey != self._tracker_key:
            self._tracker, self._tracker_key = self._make_tracker(spec), tkey
        elif not persist:
            self._tracker.reset()
            if self._gmc is not None:
                self._gmc.reset_params()
        kwargs = dict(kwargs)
        kwargs["conf"] = kwargs.get("conf") or 0.1      # ultralytics Model.track default
        self._obj_feats = bool(getattr(self._tracker, "with_reid", False))   # `with_reid: true, model: auto`: the detector keeps a vector per box
        if self._obj_feats and self.is_rtdetr:
            raise NotImplementedError("with_reid: true, model: auto needs the YOLOv8 Detect layer's inputs; not implemented for RT-DETR")
        frame = np.ascontiguousarray(source, dtype=np.uint8)
        d = self._detector(frame.shape[:2], kwargs).detect(frame)
        res = Results(Boxes(d.xyxy, d.conf, d.cls, None), d.speed, frame.shape[:2], self.names)
        b = res.boxes
        # ultralytics' on_predict_postprocess_end (trackers/track.py, pinned >= 8.4.80) calls tracker.update(det, img) on EVERY
        # frame, detections or not: the frame counter advances, unmatched tracks go lost / age out and BoT-SORT's GMC moves
        # its previous frame on. The engine follows the same rule (engine.py "Ordering rules").
        warp = None
        if self._gmc_method is not None:                    # BOTSORT.update: camera motion first, on this frame's gray image
            frame = np.asarray(source)
            if self._gmc is None or self._gmc.frame_hw != tuple(frame.shape[:2]):
                from .gmc import make_gmc

                self._gmc = make_gmc(frame.shape[:2], method=self._gmc_method, ctx=self.ctx)
            g = self._det.gray_dptr(0) if self._det is not None else (0, 0, 0)
            if getattr(self._gmc, "wants_frames", False):   # ecc: the blur comes before the reduction, it works on the frame itself
                warp = self._gmc.apply(frame)
            elif g[0] and (g[1], g[2]) == (frame.shape[0] // 2, frame.shape[1] // 2):
                self._gmc.submit_gray_dev(g[0], g[1], g[2])  # the half-resolution gray the detector left in HBM
                warp = self._gmc.collect()
            else:
                warp = self._gmc.apply(frame)
        xyxy, ids, score, cls, _idx = self._tracker.update(b._xyxy, b._conf, b._cls.astype(np.int32), gmc=warp, feats=d.feats)
        if len(ids) == 0:                                   # the raw detections (or no rows) with id None (extract.py:161-165)
            return [res]
        res.boxes = Boxes(xyxy, score, cls, ids)
        return [res]

    @property
    def detector(self) -> Detector | None:
        return self._det


class RTDETR(YOLO):
    """ultralytics.RTDETR(model): what the reference constructs when the model's yaml names RT-DETR (extract.py:224-225). The
    detector family is read off the tensors either way; this class only refuses a YOLOv8 graph."""

    def __init__(self, model, task: str = "detect", ctx: _lib.Context | None = None):
        super().__init__(model, task, ctx)
        if not self.is_rtdetr:
            raise ValueError("RTDETR(model): the tensors describe a YOLOv8 graph; use YOLO(model)")
