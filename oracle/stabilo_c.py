"""ORACLE -- test infrastructure, not product code.

ctypes face of oracle/stabilo_ref.c, the plain-C restatement of oracle/stabilo_ref.py that bench.py's CPU baseline times
on one thread and on all host cores. Same call sequence as StabilizerRef. The BRIEF table and
the foreground rectangles come from the numpy oracle (brief_pattern, mask_rects): one definition of each.
`build()` compiles the library with gcc (-O2 -fopenmp) into oracle/_build/; only tests/, __graft_entry__ and the bench's
cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from .stabilo_ref import brief_pattern, mask_rects

_HERE = Path(__file__).resolve().parent
SRC, LIB = _HERE / "stabilo_ref.c", _HERE / "_build" / "libstabilo_ref.so"
_lib = None


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", str(LIB), str(SRC), "-lm"], check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.stab_extract.restype = C.c_int
        _lib.stab_match.restype = C.c_int
        _lib.stab_ransac.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n: int) -> None:
    lib().stab_set_threads(int(n))


def gray(frame_bgr: np.ndarray, half: bool) -> np.ndarray:
    f = np.ascontiguousarray(frame_bgr, np.uint8)
    h, w = f.shape[:2]
    out = np.empty((h // 2, w // 2) if half else (h, w), np.uint8)
    lib().stab_gray(_p(f), h, w, int(half), _p(out))
    return out


def extract(g: np.ndarray, boxes_xywh, cfg: dict, max_features: int, pattern: np.ndarray) -> dict:
    gh, gw = g.shape
    rects = np.zeros((0, 4), np.int32)
    if cfg.get("mask_use", True) and boxes_xywh is not None and len(boxes_xywh):
        rects = np.asarray(mask_rects(boxes_xywh, cfg["downsample_ratio"], cfg["mask_margin_ratio"], gw, gh), np.int32).reshape(-1, 4)
    cap = max_features + 64
    xy, level, bin_ = np.zeros((cap, 2), np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    desc, px = np.zeros((cap, 32), np.uint8), np.zeros((cap, 2), np.int32)
    pat = np.ascontiguousarray(pattern, np.int8)
    k = lib().stab_extract(_p(np.ascontiguousarray(g)), gh, gw, _p(np.ascontiguousarray(rects)), len(rects), int(cfg["n_levels"]),
                           C.c_float(cfg["scale_factor"]), int(cfg["fast_threshold"]), C.c_float(cfg["downsample_ratio"]), int(max_features),
                           _p(pat), cap, _p(xy), _p(level), _p(bin_), _p(desc), _p(px))
    if k < 0:
        raise RuntimeError(f"stab_extract failed ({k})")
    return dict(xy=xy[:k], level=level[:k], bin=bin_[:k], desc=desc[:k], px=px[:k])


def match(dq: np.ndarray, dt: np.ndarray, ratio: float, keep_all: bool = False):
    nq, nt = len(dq), len(dt)
    qi, ti, di = np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32), np.zeros(max(nq, 1), np.int32)
    n = lib().stab_match(_p(np.ascontiguousarray(dq)), nq, _p(np.ascontiguousarray(dt)), nt, C.c_float(ratio), int(keep_all), _p(qi), _p(ti), _p(di))
    return qi[:n], ti[:n], di[:n]


def ransac_homography(pts_q, pts_t, frame_wh, thr: float, n_hyp: int, seed: int, affine: bool = False):
    pq, pt = np.ascontiguousarray(pts_q, np.float32), np.ascontiguousarray(pts_t, np.float32)
    H, n_inl = np.zeros(9, np.float64), C.c_int(0)
    ok = lib().stab_ransac(_p(pq), _p(pt), len(pq), int(frame_wh[0]), int(frame_wh[1]), C.c_float(thr), int(n_hyp), C.c_uint32(seed & 0xFFFFFFFF), int(affine),
                           _p(H), C.byref(n_inl))
    return (H.reshape(3, 3), n_inl.value) if ok else (None, 0)


class StabilizerC:
    """StabilizerRef's call sequence over the C restatement."""

    def __init__(self, cfg: dict, frame_hw, pattern: np.ndarray | None = None, n_hyp: int = 2048):
        self.cfg, self.hw, self.n_hyp = cfg, frame_hw, n_hyp
        self.pattern = brief_pattern() if pattern is None else pattern
        self.half = cfg["downsample_ratio"] == 0.5
        self.ref = self.cur = None
        if cfg.get("clahe"):
            raise NotImplementedError("the C restatement has no CLAHE stage (the numpy oracle has)")

    def set_ref_frame(self, frame_bgr, boxes=None):
        n_ref = int(np.floor(self.cfg["max_features"] * self.cfg["ref_multiplier"] + 0.5))
        self.ref = extract(gray(frame_bgr, self.half), boxes, self.cfg, n_ref, self.pattern)

    def stabilize(self, frame_bgr, boxes=None):
        self.cur = extract(gray(frame_bgr, self.half), boxes, self.cfg, self.cfg["max_features"], self.pattern)
        self.m = match(self.cur["desc"], self.ref["desc"], self.cfg["filter_ratio"], keep_all=self.cfg.get("filter_type") == "none")
        qi, ti, _ = self.m
        return ransac_homography(self.cur["xy"][qi], self.ref["xy"][ti], (self.hw[1], self.hw[0]), self.cfg["ransac_threshold"], self.n_hyp,
                                 self.cfg["seed"], affine=self.cfg.get("transformation_type") == "affine")
