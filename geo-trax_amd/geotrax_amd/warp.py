"""Perspective frame warp over the C ABI (gtx_warp_frame / gtx_warp_frame_dev).

Reference behaviour replaced: ``cv2.warpPerspective(frame, transforms[frame_num], (w, h))`` in the visualisation
modes 1 and 4 (geotrax/visualize.py:285-289) -- the one place the reference resamples whole frames. BGR uint8,
bilinear, constant-0 border, OpenCV's 1/32-pixel coordinate quantisation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr


def warp_perspective(frame: np.ndarray, H: np.ndarray, ctx: _lib.Context | None = None) -> np.ndarray:
    """cv2.warpPerspective(frame, H, (w, h)) for a host BGR frame; returns a new host frame."""
    ctx = ctx or _lib.default_context()
    f = np.ascontiguousarray(frame, dtype=np.uint8)
    if f.ndim != 3 or f.shape[2] != 3:
        raise ValueError(f"expected an [h, w, 3] uint8 frame, got {frame.shape}")
    Hm = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    out = np.empty_like(f)
    check(ctx.lib.gtx_warp_frame(ctx.handle, ptr(f), f.shape[0], f.shape[1], ptr(Hm), ptr(out)))
    return out


class FrameWarper:
    """Warps a stream of equally sized frames through two persistent HBM buffers (one upload, one kernel, one
    download per frame; the visualisation loop's shape). `warp_dev` works on device pointers only."""

    def __init__(self, frame_hw: tuple[int, int], ctx: _lib.Context | None = None):
        self.ctx = ctx or _lib.default_context()
        self.h, self.w = int(frame_hw[0]), int(frame_hw[1])
        self.nbytes = self.h * self.w * 3
        self.src = self.ctx.dev_alloc(self.nbytes)
        self.dst = self.ctx.dev_alloc(self.nbytes)

    def close(self):
        for name in ("src", "dst"):
            p = getattr(self, name, None)
            if p:
                self.ctx.dev_free(p)
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def warp_dev(self, src_dptr: int, H: np.ndarray, dst_dptr: int) -> None:
        """Asynchronous: enqueued on the context's stream."""
        Hm = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
        check(self.ctx.lib.gtx_warp_frame_dev(self.ctx.handle, C.c_void_p(src_dptr), self.h, self.w, ptr(Hm), C.c_void_p(dst_dptr)))

    def __call__(self, frame: np.ndarray, H: np.ndarray | None) -> np.ndarray:
        """frame -> stabilized frame; H None (the reference frame / an unregistered frame) returns the frame unchanged,
        as visualize.py:285 does when the frame number has no transform."""
        if H is None:
            return frame
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        if f.shape != (self.h, self.w, 3):
            raise ValueError(f"frame is {f.shape}, warper was built for {(self.h, self.w, 3)}")
        self.ctx.dev_upload(self.src, f)
        self.warp_dev(self.src, H, self.dst)
        out = np.empty_like(f)
        self.ctx.dev_download(out, self.dst)            # stream-ordered behind the kernel
        return out
