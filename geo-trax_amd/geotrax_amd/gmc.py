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
        if method in ("orb", "sift"):
            raise ValueError(f"GMC(method='{method}'): use gmc.make_gmc (FeatureGMC)")
        if method != "sparseOptFlow":
            raise ValueError(f"GMC(method='{method}'): use gmc.make_gmc ('sparseOptFlow', 'orb', 'sift' and 'ecc' are implemented)")
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


def estimate_affine_partial(p_xy: np.ndarray, q_xy: np.ndarray, seed: int = 0):
    """cv2.estimateAffinePartial2D(p, q, RANSAC) stand-in (gtx_op_estimate_affine_partial, host C++): 2x3 float64 or None."""
    p = np.ascontiguousarray(p_xy, np.float32).reshape(-1, 2)
    q = np.ascontiguousarray(q_xy, np.float32).reshape(-1, 2)
    A, valid, inl = np.zeros(6, np.float64), C.c_int(), C.c_int()
    check(_lib.load().gtx_op_estimate_affine_partial(ptr(p) if len(p) else None, ptr(q) if len(q) else None, len(p), seed, ptr(A), C.byref(valid), C.byref(inl)))
    return (A.reshape(2, 3), int(inl.value)) if valid.value else (None, 0)


def filter_matches(prev_xy: np.ndarray, cur_xy: np.ndarray, frame_hw) -> np.ndarray:
    """ultralytics GMC.apply_features' two spatial filters on ratio-tested matches: |displacement| < 0.25 x (width, height), then
    displacement - mean < 2.5 x std per axis (one-sided, as upstream writes it). -> boolean mask of the kept matches."""
    d = prev_xy.astype(np.float64) - cur_xy.astype(np.float64)
    keep = (np.abs(d[:, 0]) < 0.25 * frame_hw[1]) & (np.abs(d[:, 1]) < 0.25 * frame_hw[0])
    if keep.sum() == 0:
        return keep
    dk = d[keep]
    inl = ((dk - dk.mean(0)) < 2.5 * dk.std(0)).all(1)
    out = np.zeros(len(d), bool)
    out[np.flatnonzero(keep)[inl]] = True
    return out


class FeatureGMC:
    """ultralytics GMC(method='orb' | 'sift', downscale=2) (`gmc_method: orb` / `sift`, default.yaml:374,419,467): keypoints and
    descriptors of the half-resolution gray image, brute-force 2-NN matching against the previous frame's with Lowe's ratio 0.9,
    the two spatial filters of apply_features, then the RANSAC partial-affine fit. Built from what the path already has:

    * orb: the stabilizer's detector / descriptor / Hamming matcher kernels (csrc/stabilizer.hip: FAST 20 on an 8-level pyramid,
      Harris ranking, 256-bit rotated BRIEF; 1000 keypoints per frame), the previous frame set as its reference each step;
    * sift: csrc/sift.hip's SIFT (plain, not RootSIFT) + the L2 2-NN kernel (csrc/match_l2.hip);
    * the fit: gtx_op_estimate_affine_partial (the sparseOptFlow fit's procedure on the host).

    Stated differences from the OpenCV calls upstream makes: the matcher's query is the CURRENT frame (stabilo's rule; upstream
    queries with the previous one), no detection-box mask, SIFT with OpenCV's default thresholds (upstream: contrast 0.02, edge 20).
    Synchronous: a frame's warp is computed when it is collected. Interface of GMC above."""

    def __init__(self, frame_hw: tuple[int, int], method: str = "orb", downscale: int = 2, seed: int = 0, ctx: _lib.Context | None = None,
                 max_features: int = 1000):
        if method not in ("orb", "sift"):
            raise NotImplementedError(f"gmc_method='{method}'")
        if downscale != 2:
            raise NotImplementedError("only downscale=2 (ultralytics' default) is implemented")
        self.method, self.seed = method, seed
        self.ctx = ctx or _lib.default_context()
        self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
        self.gh, self.gw = self.frame_hw[0] // 2, self.frame_hw[1] // 2
        self.max_features = int(max_features)
        self.stats = np.zeros(3, np.int32)              # keypoints of the previous frame, matches kept, inliers
        self.valid = False
        self._pending = []                               # gray images submitted and not yet collected (device pointers or host arrays)
        self._prev = None
        if method == "orb":
            from .stabilizer import Stabilizer

            self._st = Stabilizer(self.frame_hw, max_features=self.max_features, ref_multiplier=1.0, filter_ratio=0.9, mask_use=False,
                                  downsample_ratio=0.5, seed=seed, ctx=self.ctx)
            self._dev = None                             # device copies of (previous, current) gray for host-side submissions
        else:
            from .registration import Sift

            self._sift = Sift((self.gh, self.gw), ctx=self.ctx)

    def close(self):
        if getattr(self, "_st", None) is not None:
            self._st.close()
            self._st = None
        if getattr(self, "_sift", None) is not None:
            self._sift.close()
            self._sift = None
        if getattr(self, "_dev", None):
            for d in self._dev:
                self.ctx.dev_free(d)
            self._dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_params(self):
        self._prev = None
        self._pending.clear()

    reset_sequence = reset_params

    # ---- submission: half-resolution gray images, resident in HBM (the detector's) or on the host
    def submit_gray_dev(self, gray_dptr: int, gh: int, gw: int) -> None:
        if (gh, gw) != (self.gh, self.gw):
            raise ValueError(f"gray image is {gw}x{gh}, the GMC was made for {self.gw}x{self.gh}")
        self._pending.append(("dev", int(gray_dptr)))

    def submit_gray(self, gray: np.ndarray) -> None:
        g = np.ascontiguousarray(gray, np.uint8)
        if g.shape != (self.gh, self.gw):
            raise ValueError(f"gray image is {g.shape[1]}x{g.shape[0]}, the GMC was made for {self.gw}x{self.gh}")
        self._pending.append(("host", g))

    def submit_frame_dev(self, frame_dptr: int, h: int, w: int, restart: bool = False) -> None:
        raise NotImplementedError("gmc_method orb / sift: a frame-sharded run primes its GMC with a BGR frame in HBM, which only 'sparseOptFlow' takes")

    def apply(self, raw_frame: np.ndarray, detections=None) -> np.ndarray:
        """2x3 float64 warp previous -> current frame (identity for the first frame). raw_frame: BGR [h, w, 3] uint8."""
        f = np.asarray(raw_frame, np.uint8).astype(np.int32)
        g = (f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14          # cv2.cvtColor(BGR2GRAY), then the exact 2x reduction
        g = ((g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        self.submit_gray(g)
        return self.collect()

    # ---- one step
    def _pairs_orb(self, kind, cur):
        n = self.gh * self.gw
        if kind == "host":                               # the stabilizer's gray entry points take device images
            if self._dev is None:
                self._dev = [self.ctx.dev_alloc(n), self.ctx.dev_alloc(n)]
            self._dev.reverse()                          # [previous, current]
            self.ctx.dev_upload(self._dev[1], cur)
            cur_ptr = self._dev[1]
        else:
            cur_ptr = cur
        prev_ptr, self._prev = self._prev, cur_ptr
        if prev_ptr is None:
            self._st.set_ref_gray_dev(cur_ptr, self.gh, self.gw)     # the sequence's first frame: its features wait as the reference
            return None
        self._st.stabilize_gray_dev(cur_ptr, self.gh, self.gw)      # features of this frame + matches against the previous frame's
        qi, ti, _ = self._st.matches()
        ref, cur_k = self._st.keypoints("ref")["xy"], self._st.keypoints("cur")["xy"]
        self._st.promote_cur()                                      # ... which now become the reference of the next frame (no second extraction)
        self.stats[0] = len(ref)
        return ref[ti], cur_k[qi]                        # previous-frame points, current-frame points (full-resolution pixels)

    def _pairs_sift(self, kind, cur):
        from .ops import match_2nn

        if kind == "dev":
            g = np.zeros((self.gh, self.gw), np.uint8)
            self.ctx.dev_download(g, cur)
        else:
            g = cur
        k = self._sift.detect_and_compute(np.repeat(g[:, :, None], 3, 2), max_features=self.max_features, root=False)
        cur_f = (k["xy"].astype(np.float32) * 2.0, k["desc"])          # full-resolution pixels
        prev_f, self._prev = self._prev, cur_f
        if prev_f is None or len(prev_f[0]) < 2 or len(cur_f[0]) == 0:
            return None
        self.stats[0] = len(prev_f[0])
        i1, i2, d1, d2 = match_2nn(cur_f[1], prev_f[1], ctx=self.ctx)
        good = (i1 >= 0) & (i2 >= 0) & (d1 < np.float32(0.9) * d2)
        return prev_f[0][i1[good]], cur_f[0][good]

    def collect(self) -> np.ndarray:
        if not self._pending:
            raise RuntimeError("GMC.collect without a submitted frame")
        kind, cur = self._pending.pop(0)
        H = np.eye(2, 3)
        self.valid = False
        self.stats[:] = 0
        pairs = self._pairs_orb(kind, cur) if self.method == "orb" else self._pairs_sift(kind, cur)
        if pairs is None or len(pairs[0]) == 0:
            return H
        prev_xy, cur_xy = pairs
        keep = filter_matches(prev_xy, cur_xy, self.frame_hw)
        self.stats[1] = int(keep.sum())
        if keep.sum() > 4:
            M, inl = estimate_affine_partial(prev_xy[keep], cur_xy[keep], self.seed)
            if M is not None:
                H, self.valid = M, True
                self.stats[2] = inl
        return H


class EccGMC:
    """ultralytics GMC(method='ecc', downscale=2) (`gmc_method: ecc`, default.yaml:374,419,467): cv2.findTransformECC with the
    Euclidean motion model, 5000 iterations / 1e-6, on cvtColor -> GaussianBlur(3x3, 1.5) -> resize(1/2) of the frame
    (csrc/ecc.hip over gtx_ecc_*; oracle/ecc_ref.py states the procedure and the OpenCV source it follows). Two properties of the
    upstream method are kept because the reference computes them: every frame is registered against the FIRST frame since
    reset_params() -- upstream never replaces its `prevFrame` for this method --, and the translation stays in half-resolution
    pixels. `last` holds {iters, status, rho} of the frame collected last (status 1 / 2: the two conditions under which
    cv2.findTransformECC raises; upstream catches it and keeps the matrix as the failed call left it, as collect() does).
    `warp`: which of OpenCV's two warpAffine forms takes the bilinear samples -- 'exact' (default; OpenCV >= 4.11, floating-point source
    positions: fits end after 5-30 iterations) or 'fixed' (through 4.10: positions rounded to 1/32 pixel, under which the coefficient
    can dither in its sixth decimal for good and the fit runs to the 5000-iteration cap).
    `replace_template=True` is NOT upstream's behaviour: every collected frame becomes the template of the next one (frame-to-frame
    warps, what a tracker's "previous -> current" compensation expects); off unless a caller asks for it.

    The method needs the BGR frame, not the detector's gray image (the blur comes before the reduction): `wants_frames` tells
    the engine to hand it every frame of a batch when the batch is submitted to a detector, on that detector's stream."""

    wants_frames = True

    def __init__(self, frame_hw: tuple[int, int], method: str = "ecc", downscale: int = 2, seed: int = 0, ctx: _lib.Context | None = None,
                 max_iters: int = 5000, eps: float = 1e-6, replace_template: bool = False, warp: str = "exact"):
        if method != "ecc":
            raise ValueError(f"EccGMC(method='{method}')")
        if downscale != 2:
            raise NotImplementedError("only downscale=2 (ultralytics' default) is implemented")
        self.ctx = ctx or _lib.default_context()
        self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
        h = C.c_void_p()
        check(self.ctx.lib.gtx_ecc_create(self.ctx.handle, self.frame_hw[0], self.frame_hw[1], int(max_iters), float(eps), C.byref(h)))
        self.handle = h
        if warp not in ("exact", "fixed"):
            raise ValueError(f"EccGMC(warp='{warp}'): 'exact' (OpenCV >= 4.11's warpAffine) or 'fixed' (through 4.10)")
        if warp == "fixed":
            check(self.ctx.lib.gtx_ecc_exact_positions(h, 0))
        if replace_template:                             # NOT upstream's behaviour: frame-to-frame warps (each collected frame becomes the next template)
            check(self.ctx.lib.gtx_ecc_replace_template(h, 1))
        self.method = "ecc"
        self.valid = False
        self.stats = np.zeros(3, np.int32)               # {iterations, status, 0}: the slot the other methods fill with point counts
        self.last = {}

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.gtx_ecc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_params(self):
        check(self.ctx.lib.gtx_ecc_reset(self.handle))

    reset_sequence = reset_params

    def submit_frame_dev(self, frame_dptr: int, h: int, w: int, restart: bool = False, producer: _lib.Context | None = None) -> None:
        if restart:
            raise NotImplementedError("gmc_method ecc registers against the first frame of the sequence: a frame-sharded run cannot prime it with another frame")
        check(self.ctx.lib.gtx_ecc_submit_dev(self.handle, producer.handle if producer is not None else None, C.c_void_p(int(frame_dptr)), int(h), int(w)))

    def submit_gray_dev(self, gray_dptr: int, gh: int, gw: int) -> None:
        raise NotImplementedError("gmc_method ecc blurs the full-resolution gray image before reducing it: it takes BGR frames (submit_frame_dev / apply)")

    def submit_frame(self, frame: np.ndarray) -> None:
        f = np.ascontiguousarray(frame, np.uint8)
        check(self.ctx.lib.gtx_ecc_submit(self.handle, ptr(f), f.shape[0], f.shape[1]))

    def collect(self) -> np.ndarray:
        A, info, rho = np.zeros(6, np.float64), np.zeros(2, np.int32), C.c_double()
        check(self.ctx.lib.gtx_ecc_collect(self.handle, ptr(A), ptr(info), C.byref(rho)))
        self.last = dict(iters=int(info[0]), status=int(info[1]), rho=float(rho.value))
        self.stats[:2] = info
        self.valid = info[0] > 0 and info[1] == 0
        return A.reshape(2, 3)

    def apply(self, raw_frame: np.ndarray, detections=None) -> np.ndarray:
        """2x3 float64 warp first frame -> this frame in half-resolution pixels (identity for the first frame)."""
        self.submit_frame(raw_frame)
        return self.collect()

    def image(self, which: int = 0) -> np.ndarray:
        out = np.zeros((self.frame_hw[0] // 2, self.frame_hw[1] // 2), np.float32)
        check(self.ctx.lib.gtx_ecc_image(self.handle, int(which), ptr(out)))
        return out


def make_gmc(frame_hw, method: str = "sparseOptFlow", **kw):
    """The GMC object of `gmc_method` (default.yaml:374): sparseOptFlow -> GMC, orb / sift -> FeatureGMC, ecc -> EccGMC."""
    if method in ("orb", "sift"):
        return FeatureGMC(frame_hw, method=method, **kw)
    if method == "ecc":
        return EccGMC(frame_hw, method=method, **kw)
    return GMC(frame_hw, method=method, **kw)
