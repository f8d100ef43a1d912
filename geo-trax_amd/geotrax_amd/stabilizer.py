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
        if detector_name not in ("orb", "sift", "rsift"):
            raise NotImplementedError(f"detector_name='{detector_name}': 'orb', 'sift' and 'rsift' are implemented (not brisk / kaze / akaze)")
        self._sift = detector_name if detector_name != "orb" else None
        if self._sift and not unused.get("sift_enable_precise_upscale", False):
            logger.warning(f"detector_name='{detector_name}': this build's SIFT doubles the base image with OpenCV's precise (half-pixel aligned) "
                           "upscaling; `sift_enable_precise_upscale: false` (default.yaml:112) is not implemented -- keypoints sit ~0.25 px from where "
                           "stabilo's default would put them, on both frames alike")
        if self._sift and (filter_type != "ratio" or transformation_type != "projective" or clahe):
            raise NotImplementedError(f"detector_name='{detector_name}' is built with filter_type 'ratio', transformation_type 'projective' and clahe off")
        if matcher_name == "flann":
            # FLANN returns APPROXIMATE nearest neighbours (LSH for binary descriptors, kd-trees for SIFT); the GPU matcher is exact
            # and faster than either here, so the option runs on it: the matches are the ones FLANN approximates, not FLANN's own
            logger.warning("matcher_name='flann': matched with the exact brute-force kernel (what FLANN approximates); a stabilo run with FLANN "
                           "may keep slightly different matches")
        elif matcher_name != "bf":
            raise NotImplementedError(f"matcher_name='{matcher_name}': 'bf' (exact brute force) and 'flann' (served by the same exact kernel) are implemented")
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
        self._ref_img = None
        self._mask_warned = False
        if frame_hw is not None and not self._sift:
            self._create(frame_hw)
        elif frame_hw is not None:
            self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
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

    # ---- detector_name sift / rsift (default.yaml:109): the registration stage's kernels (csrc/sift.hip, match_l2.hip, the RANSAC of
    # stabilizer.hip behind gtx_register_images) on the working-resolution frames; one blocking call per frame, host frames only
    def _working(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        r = float(self._kw["downsample_ratio"])
        if r == 1.0:
            return f
        if r != 0.5:
            raise NotImplementedError("detector_name sift / rsift: downsample_ratio 1.0 or 0.5")
        h, w = f.shape[0] // 2 * 2, f.shape[1] // 2 * 2
        q = f[:h, :w].astype(np.uint16)
        return ((q[0::2, 0::2] + q[0::2, 1::2] + q[1::2, 0::2] + q[1::2, 1::2] + 2) >> 2).astype(np.uint8)      # cv2.resize(INTER_LINEAR) at exactly 1/2

    def _sift_register(self, frame, boxes):
        from .registration import register_once

        if boxes is not None and len(boxes) and self._kw["mask_use"] and not self._mask_warned:
            self._mask_warned = True
            logger.warning(f"detector_name='{self._sift}': the vehicle mask (mask_use) is not applied on this path; keypoints on vehicles reach the matcher")
        cur = self._working(frame)
        n_cur = int(self._kw["max_features"])
        Hh, stats, _ = register_once(cur, self._ref_img, max_features=max(n_cur, 4), filter_ratio=self._kw["filter_ratio"],
                                     ransac_epipolar_threshold=self._kw["ransac_threshold"], ransac_max_iter=self._kw["ransac_max_iter"],
                                     ransac_confidence=self._kw["ransac_confidence"], rsift_eps=1e-8 if self._sift == "rsift" else -1.0,
                                     seed=self._kw["seed"], ctx=self.ctx)
        self._stats[:] = (stats[1], stats[0], stats[2], stats[3])          # (reference, current, matches, inliers)
        H = np.zeros(9, np.float64)
        valid = C.c_int(0)
        if Hh is not None:
            r = float(self._kw["downsample_ratio"])
            D = np.diag([r, r, 1.0])
            Hf = np.linalg.inv(D) @ Hh @ D                                   # working-resolution pixels -> frame pixels on both sides
            H[:] = (Hf / Hf[2, 2]).ravel()
            valid = C.c_int(1)
        b, _ = self._boxes(boxes)
        self._finish(H, valid, b)

    # ---- stabilo interface
    def set_ref_frame(self, frame: np.ndarray, boxes: np.ndarray | None = None) -> None:
        if self._sift:
            self._ref_img = self._working(frame)
            self.frame_hw = tuple(np.asarray(frame).shape[:2])
            self._H = self._H_raw = self._H_last_known = None
            self._cur_boxes = None
            return
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        if self.handle is None:
            self._create(f.shape[:2])
        b, n = self._boxes(boxes)
        check(self.ctx.lib.gtx_stabilizer_set_ref_frame(self.handle, ptr(f), f.shape[0], f.shape[1], ptr(b), n))
        self._H = self._H_raw = self._H_last_known = None
        self._cur_boxes = None

    def set_ref_gray_dev(self, gray_dptr: int, gh: int, gw: int, boxes=None) -> None:
        if self._sift:
            raise NotImplementedError(f"detector_name='{self._sift}' takes host frames (set_ref_frame / stabilize): run with engine: {{pipelined: false}}")
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
        if self._sift:
            if self._ref_img is None:
                raise RuntimeError("stabilize() before set_ref_frame()")
            return self._sift_register(frame, boxes)
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
    def promote_cur(self) -> None:
        """The frame stabilized last becomes the reference (its features are kept, nothing is extracted again); ref_multiplier 1 only."""
        check(self.ctx.lib.gtx_stabilizer_promote_cur(self.handle))

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
