"""Image registration over the C ABI (gtx_register_images, gtx_sift_*).

Reference behaviour replaced: geotrax/utils/registration.py:21-95 (`estimate_homography`): the
georeference stage registers a video frame / master frame / orthophoto cut-out pair through a stabilo
Stabilizer configured for RootSIFT + brute-force matching + ratio filter + USAC_MAGSAC, and retries
with half the features when no model comes back. Same function name, keywords, return tuple and
retry rule here; the arithmetic runs on the GPU (csrc/sift.hip, match_l2.hip, stabilizer.hip).
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib
from ._lib import RegConfig, check, ptr

USAC_MAGSAC = 38          # cv2.USAC_MAGSAC, the reference's default ransac_method


def register_once(img_src: np.ndarray, img_dst: np.ndarray, *, max_features: int, filter_ratio: float, ransac_epipolar_threshold: float,
                  ransac_max_iter: int, ransac_confidence: float, rsift_eps: float, seed: int = 0, ctx: _lib.Context | None = None):
    """One registration attempt -> (H or None, stats [n_src, n_dst, n_matches, n_inliers], timings ms)."""
    ctx = ctx or _lib.default_context()
    src = np.ascontiguousarray(img_src, np.uint8)
    dst = np.ascontiguousarray(img_dst, np.uint8)
    if src.ndim != 3 or src.shape[2] != 3 or dst.ndim != 3 or dst.shape[2] != 3:
        raise ValueError("registration expects BGR uint8 images [h, w, 3]")
    cfg = RegConfig(max_features=int(max_features), filter_ratio=float(filter_ratio), ransac_threshold=float(ransac_epipolar_threshold),
                    ransac_max_iter=int(ransac_max_iter), ransac_confidence=float(ransac_confidence), rsift_eps=float(rsift_eps), seed=seed)
    H, valid = np.zeros(9, np.float64), C.c_int()
    stats, tm = np.zeros(4, np.int32), np.zeros(4, np.float32)
    check(ctx.lib.gtx_register_images(ctx.handle, C.byref(cfg), ptr(src), src.shape[0], src.shape[1], ptr(dst), dst.shape[0], dst.shape[1],
                                      ptr(H), C.byref(valid), ptr(stats), ptr(tm)))
    return (H.reshape(3, 3) if valid.value else None), stats, tm


def estimate_homography(img_src: np.ndarray, img_dst: np.ndarray, logger: logging.Logger, *, detector_name: str = 'rsift',
                        matcher_name: str = 'bf', filter_type: str = 'ratio', sift_enable_precise_upscale: bool = True,
                        max_features: int = 250000, filter_ratio: float = 0.55, ransac_method: int = USAC_MAGSAC,
                        ransac_epipolar_threshold: float = 3.0, ransac_max_iter: int = 10000, ransac_confidence: float = 0.999999,
                        rsift_eps: float = 1e-8, ctx: _lib.Context | None = None) -> tuple:
    """H mapping source -> destination image coordinates (registration.py:21-95).

    Returns (H, inliers_count, num_matches, (n_src_kpts, n_dst_kpts)) or (None, None, None, None).
    If detection or matching fails, `max_features` is halved and retried (down to >10000)."""
    if detector_name != 'rsift':
        raise NotImplementedError(f"detector_name='{detector_name}': only 'rsift' is implemented for registration")
    if matcher_name != 'bf' or filter_type != 'ratio':
        raise NotImplementedError("only matcher_name='bf' with filter_type='ratio' is implemented")
    if not sift_enable_precise_upscale:
        raise NotImplementedError("sift_enable_precise_upscale=False is not implemented")
    max_features_to_try = max_features
    while max_features_to_try > 10000:
        H, stats, _ = register_once(img_src, img_dst, max_features=max_features_to_try, filter_ratio=filter_ratio,
                                    ransac_epipolar_threshold=ransac_epipolar_threshold, ransac_max_iter=ransac_max_iter,
                                    ransac_confidence=ransac_confidence, rsift_eps=rsift_eps, ctx=ctx)
        if H is not None:
            return H, int(stats[3]), int(stats[2]), (int(stats[0]), int(stats[1]))
        max_features_to_try //= 2
        logger.warning(f"Feature detection or matching failed with {max_features_to_try * 2} max_features. "
                       f"Trying with {max_features_to_try} max_features.")
    logger.error("Feature detection failed with all attempted feature counts.")
    return None, None, None, None


class Sift:
    """The detector stage on its own (parity tests): RootSIFT keypoints + descriptors of one image."""

    def __init__(self, max_hw: tuple[int, int], ctx: _lib.Context | None = None):
        self.ctx = ctx or _lib.default_context()
        h = C.c_void_p()
        check(self.ctx.lib.gtx_sift_create(self.ctx.handle, int(max_hw[0]), int(max_hw[1]), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.gtx_sift_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect_and_compute(self, img_bgr: np.ndarray, max_features: int = 250000, root: bool = True, eps: float = 1e-8, cap: int = 1 << 18):
        img = np.ascontiguousarray(img_bgr, np.uint8)
        n = C.c_int()
        kp, octv, desc = np.zeros((cap, 5), np.float32), np.zeros(cap, np.int32), np.zeros((cap, 128), np.float32)
        check(self.ctx.lib.gtx_sift_detect(self.handle, ptr(img), img.shape[0], img.shape[1], max_features, int(root), eps, cap, C.byref(n),
                                           ptr(kp), ptr(octv), ptr(desc)))
        k = min(n.value, cap)
        return dict(xy=kp[:k, :2].copy(), size=kp[:k, 2].copy(), angle=kp[:k, 3].copy(), response=kp[:k, 4].copy(),
                    octave=octv[:k].copy(), desc=desc[:k].copy(), count=int(n.value))

    def stage_ms(self) -> dict:
        """GPU time of the last detect_and_compute by stage, and the pixel count of the doubled base image."""
        out = np.zeros(4, np.float32)
        check(self.ctx.lib.gtx_sift_stage_ms(self.handle, ptr(out)))
        return dict(pyramid=float(out[0]), keypoints=float(out[1]), describe=float(out[2]), base_pixels=float(out[3]))

    def pyramid(self, kind: int, octave: int, layer: int) -> np.ndarray:
        h, w, no = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_sift_pyramid(self.handle, kind, octave, layer, 0, None, C.byref(h), C.byref(w), C.byref(no)))
        out = np.zeros((h.value, w.value), np.float32)
        check(self.ctx.lib.gtx_sift_pyramid(self.handle, kind, octave, layer, out.size, ptr(out), C.byref(h), C.byref(w), C.byref(no)))
        return out

    def n_octaves(self) -> int:
        h, w, no = C.c_int(), C.c_int(), C.c_int()
        check(self.ctx.lib.gtx_sift_pyramid(self.handle, 0, 0, 0, 0, None, C.byref(h), C.byref(w), C.byref(no)))
        return no.value
