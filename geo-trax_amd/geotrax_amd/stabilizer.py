"""Stabilizer object over the C ABI (gtx_stabilizer_*), with stabilo's method names.

Reference behaviour replaced: ``stabilo.Stabilizer`` as driven by geotrax/extract.py:139,177-187
(per-frame registration against the first processed frame) and geotrax/utils/registration.py:59-85
(image-to-image registration). Constructor keywords are the keys of the reference's ``stabilo:``
config block (geotrax/cfg/default.yaml:100-145); unknown or unsupported choices raise.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib, geometry
from ._lib import StabConfig, check, ptr

logger = logging.getLogger(__name__)


class Stabilizer:
    def __init__(self, frame_hw: tuple[int, int] | None = None, *, detector_name: str = "orb", matcher_name: str = "bf",
                 filter_type: str = "ratio", transformation_type: str = "projective", clahe: bool = False,
                 downsample_ratio: float = 0.5, max_features: int = 2000, ref_multiplier: float = 2.0,
                 filter_ratio: float = 0.9, ransac_method: int = 38, ransac_epipolar_threshold: float = 2.0,
                 ransac_max_iter: int = 5000, ransac_confidence: float = 0.999999, mask_use: bool = True,
                 mask_margin_ratio: float = 0.15, min_good_match_count_warning: int = 20,
                 min_inliers_match_count_warning: int = 10, fast_threshold: int = 20, n_levels: int = 8,
                 scale_factor: float = 1.2, seed: int = 0, match_query_frame: str = "current",
                 ctx: _lib.Context | None = None, **unused):
        if detector_name != "orb":
            raise NotImplementedError(f"detector_name='{detector_name}': only 'orb' is implemented on the GPU path")
        if matcher_name != "bf":
            raise NotImplementedError(f"matcher_name='{matcher_name}': only 'bf' (exact brute force) is implemented")
        if filter_type not in ("ratio", "none"):
            raise NotImplementedError(f"filter_type='{filter_type}': only 'ratio' and 'none' are implemented")
        if transformation_type not in ("projective", "affine"):
            raise ValueError(f"transformation_type='{transformation_type}' (choices: projective, affine)")
        self.ctx = ctx or _lib.default_context()
        self._kw = dict(downsample_ratio=downsample_ratio, max_features=max_features, ref_multiplier=ref_multiplier,
                        filter_ratio=filter_ratio, ransac_threshold=ransac_epipolar_threshold, ransac_max_iter=ransac_max_iter,
                        ransac_confidence=ransac_confidence, mask_use=int(mask_use), mask_margin_ratio=mask_margin_ratio,
                        fast_threshold=fast_threshold, n_levels=n_levels, scale_factor=scale_factor, seed=seed, clahe=int(bool(clahe)),
                        affine=int(transformation_type == "affine"), filter_type=int(filter_type == "none"))
        self.min_good, self.min_inl = min_good_match_count_warning, min_inliers_match_count_warning
        self.handle = None
        self.frame_hw = None
        if frame_hw is not None:
            self._create(frame_hw)
        self._H = None                   # what get_cur_trans_matrix() returns: this frame's transform, else the last known one
        self._H_raw = None               # this frame's own transform, None when the frame could not be registered
        self._H_last_known = None
        self._stats = np.zeros(4, np.int32)
        self._cur_boxes = None

    # ---- lifecycle
    def _create(self, frame_hw):
        cfg = StabConfig(frame_h=int(frame_hw[0]), frame_w=int(frame_hw[1]), **self._kw)
        h = C.c_void_p()
        check(self.ctx.lib.gtx_stabilizer_create(self.ctx.handle, C.byref(cfg), C.byref(h)))
        self.handle, self.frame_hw = h, (int(frame_hw[0]), int(frame_hw[1]))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.gtx_stabilizer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _boxes(boxes):
        if boxes is None or len(boxes) == 0:
            return None, 0
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
        return b, len(b)

    # ---- stabilo interface
    def set_ref_frame(self, frame: np.ndarray, boxes: np.ndarray | None = None) -> None:
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        if self.handle is None:
            self._create(f.shape[:2])
        b, n = self._boxes(boxes)
        check(self.ctx.lib.gtx_stabilizer_set_ref_frame(self.handle, ptr(f), f.shape[0], f.shape[1], ptr(b), n))
        self._H = self._H_raw = self._H_last_known = None
        self._cur_boxes = None

    def set_ref_gray_dev(self, gray_dptr: int, gh: int, gw: int, boxes=None) -> None:
        b, n = self._boxes(boxes)
        check(self.ctx.lib.gtx_stabilizer_set_ref_gray_dev(self.handle, C.c_void_p(gray_dptr), gh, gw, ptr(b), n))
        self._H = self._H_raw = self._H_last_known = None
        self._cur_boxes = None

    def _finish(self, H, valid, boxes):
        # stabilo keeps `trans_matrix_last_known`: a frame that cannot be registered (too few matches, no model) takes the
        # previous valid transform for its boxes and reports it as its matrix; before the first valid one there is none.
        self._H_raw = H.reshape(3, 3).copy() if valid.value else None
        if self._H_raw is not None:
            self._H_last_known = self._H_raw
        self._H = self._H_last_known
        self._cur_boxes = boxes
        if self._stats[2] < self.min_good:
            logger.warning(f"Only {int(self._stats[2])} good matches found.")
        elif self._H is not None and self._stats[3] < self.min_inl:
            logger.warning(f"Only {int(self._stats[3])} inliers found.")

    def stabilize(self, frame: np.ndarray, boxes: np.ndarray | None = None) -> None:
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        b, n = self._boxes(boxes)
        H, valid = np.zeros(9, np.float64), C.c_int()
        check(self.ctx.lib.gtx_stabilizer_stabilize(self.handle, ptr(f), f.shape[0], f.shape[1], ptr(b), n, ptr(H),
                                                    C.byref(valid), ptr(self._stats)))
        self._finish(H, valid, b)

    def stabilize_gray_dev(self, gray_dptr: int, gh: int, gw: int, boxes=None) -> None:
        b, n = self._boxes(boxes)
        H, valid = np.zeros(9, np.float64), C.c_int()
        check(self.ctx.lib.gtx_stabilizer_stabilize_gray_dev(self.handle, C.c_void_p(gray_dptr), gh, gw, ptr(b), n, ptr(H),
                                                             C.byref(valid), ptr(self._stats)))
        self._finish(H, valid, b)

    def submit_gray_dev(self, gray_dptr: int, gh: int, gw: int, boxes=None) -> None:
        """Asynchronous stabilize: enqueue on the stabilizer's stream; pair with collect()."""
        b, n = self._boxes(boxes)
        check(self.ctx.lib.gtx_stabilizer_submit_gray_dev(self.handle, C.c_void_p(gray_dptr), gh, gw, ptr(b), n))
        self._pending_boxes = b

    def collect(self) -> None:
        H, valid = np.zeros(9, np.float64), C.c_int()
        check(self.ctx.lib.gtx_stabilizer_collect(self.handle, ptr(H), C.byref(valid), ptr(self._stats)))
        self._finish(H, valid, self._pending_boxes)

    def last_ms(self) -> float:
        """GPU time (ms) of the last collected asynchronous pass."""
        ms = C.c_float()
        check(self.ctx.lib.gtx_stabilizer_last_ms(self.handle, C.byref(ms)))
        return float(ms.value)

    def get_cur_trans_matrix(self, raw: bool = False) -> np.ndarray | None:
        """3x3 float64 mapping current-frame pixels to reference-frame pixels, or None. raw=True: None also when this
        frame itself could not be registered (the engine, whose stabilizer objects take turns, keeps the last known
        transform in frame order itself)."""
        H = self._H_raw if raw else self._H
        return None if H is None else H.copy()

    @property
    def registered(self) -> bool:
        """False when the last frame took the last known transform (or none) instead of one of its own."""
        return self._H_raw is not None

    def transform_cur_boxes(self) -> np.ndarray:
        """The boxes given to the last stabilize() call, mapped into the reference frame (xywh)."""
        if self._cur_boxes is None:
            return np.zeros((0, 4), np.float32)
        if self._H is None:
            return self._cur_boxes.copy()
        return geometry.warp_boxes(self._H, self._cur_boxes)

    def get_cur_num_keypoints(self) -> tuple[int, int]:
        return int(self._stats[0]), int(self._stats[1])  # (reference, current)

    def get_cur_num_matches(self) -> int:
        return int(self._stats[2])

    def get_cur_inliers_count(self) -> int:
        return int(self._stats[3])

    # ---- introspection for the parity tests
    def keypoints(self, which: str = "cur"):
        cap = 1 << 16
        n = C.c_int()
        xy, lvl, ab = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        desc = np.zeros((cap, 32), np.uint8)
        check(self.ctx.lib.gtx_stabilizer_keypoints(self.handle, 0 if which == "ref" else 1, cap, C.byref(n), ptr(xy), ptr(lvl),
                                                    ptr(ab), ptr(desc)))
        k = n.value
        return dict(xy=xy[:k].copy(), level=lvl[:k].copy(), bin=ab[:k].copy(), desc=desc[:k].copy())

    def matches(self):
        cap = 1 << 16
        n = C.c_int()
        q, t, d = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        check(self.ctx.lib.gtx_stabilizer_matches(self.handle, cap, C.byref(n), ptr(q), ptr(t), ptr(d)))
        k = n.value
        return q[:k].copy(), t[:k].copy(), d[:k].copy()

    def pattern(self) -> np.ndarray:
        out = np.zeros((256, 256, 4), np.int8)
        check(self.ctx.lib.gtx_stabilizer_pattern(self.handle, ptr(out)))
        return out
