"""Global motion compensation object over the C ABI (gtx_gmc_*), with ultralytics' GMC interface.

Reference behaviour replaced: ultralytics.trackers.utils.gmc.GMC(method='sparseOptFlow', downscale=2)
as BOTSORT.update calls it once per frame inside model.track(..., persist=True)
(geotrax/extract.py:153, tracker block geotrax/cfg/default.yaml:362-374).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


class GMC:
    def __init__(self, frame_hw: tuple[int, int], method: str = "sparseOptFlow", downscale: int = 2, seed: int = 0,
                 ctx: _lib.Context | None = None):
        if method in (None, "none", "None"):
            raise ValueError("GMC(method='none'): do not create the object, pass no warp to the tracker")
        if method != "sparseOptFlow":
            raise NotImplementedError(f"gmc_method='{method}': only 'sparseOptFlow' is implemented on the GPU path")
        if downscale != 2:
            raise NotImplementedError("only downscale=2 (ultralytics' default) is implemented")
        self.ctx = ctx or _lib.default_context()
        h = C.c_void_p()
        check(self.ctx.lib.gtx_gmc_create(self.ctx.handle, int(frame_hw[0]), int(frame_hw[1]), seed, C.byref(h)))
        self.handle = h
        self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
        self.stats = np.zeros(3, np.int32)
        self.valid = False

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.gtx_gmc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_params(self):                                 # ultralytics' name
        check(self.ctx.lib.gtx_gmc_reset(self.handle))

    def apply(self, raw_frame: np.ndarray, detections=None) -> np.ndarray:
        """2x3 float64 warp previous -> current frame (identity for the first frame)."""
        f = np.ascontiguousarray(raw_frame, np.uint8)
        A, valid = np.zeros(6, np.float64), C.c_int()
        check(self.ctx.lib.gtx_gmc_apply(self.handle, ptr(f), f.shape[0], f.shape[1], ptr(A), C.byref(valid), ptr(self.stats)))
        self.valid = bool(valid.value)
        return A.reshape(2, 3)

    def submit_gray_dev(self, gray_dptr: int, gh: int, gw: int) -> None:
        check(self.ctx.lib.gtx_gmc_submit_gray_dev(self.handle, C.c_void_p(gray_dptr), gh, gw))

    def submit_frame_dev(self, frame_dptr: int, h: int, w: int, restart: bool = False) -> None:
        """A BGR frame in HBM; restart=True opens a new sequence with it (identity warp, next frame is compared with it)."""
        check(self.ctx.lib.gtx_gmc_submit_frame_dev(self.handle, C.c_void_p(frame_dptr), h, w, int(restart)))

    def reset_sequence(self) -> None:
        """The next submitted frame opens a new sequence (identity warp), with frames still in flight: only the
        submitting side's state changes. reset_params() is the blocking variant used between clips."""
        check(self.ctx.lib.gtx_gmc_restart(self.handle))

    def collect(self) -> np.ndarray:
        A, valid = np.zeros(6, np.float64), C.c_int()
        check(self.ctx.lib.gtx_gmc_collect(self.handle, ptr(A), C.byref(valid), ptr(self.stats)))
        self.valid = bool(valid.value)
        return A.reshape(2, 3)

    def points(self, which: int):
        n = C.c_int()
        xy, st = np.zeros((1024, 2), np.float32), np.zeros(1024, np.int32)
        check(self.ctx.lib.gtx_gmc_points(self.handle, which, 1024, C.byref(n), ptr(xy), ptr(st)))
        return xy[:n.value].copy(), st[:n.value].astype(bool)
