"""numpy-facing wrappers of the operator-level C ABI (gtx_op_*): host arrays in, host arrays out.

These exist so that every HIP kernel of the detector can be checked against oracle/ on the exact
layer shapes. Activations are NHWC; dtype float16 or float32.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ConvDesc, check, ptr

GTX_F16, GTX_F32, GTX_F32S = 0, 1, 2


def _dt(a: np.ndarray) -> int:
    if a.dtype == np.float16:
        return GTX_F16
    if a.dtype == np.float32:
        return GTX_F32
    raise TypeError(f"activations must be float16 or float32, got {a.dtype}")


def conv2d(x: np.ndarray, w_ohwi: np.ndarray, bias: np.ndarray | None = None, *, stride: int = 1,
           act: bool = True, residual: np.ndarray | None = None, in_coff: int = 0, cin: int | None = None,
           out: np.ndarray | None = None, out_coff: int = 0, split: bool = False, ctx: _lib.Context | None = None) -> np.ndarray:
    """act(conv2d(x[..., in_coff:in_coff+cin], w) + b) (+ residual), written into
    out[..., out_coff:out_coff+cout]. x: [n,h,w,cs]; w_ohwi: [cout,k,k,cin] fp32. split=True (float32 arrays only):
    the split-f16x3 kernel (GTX_F32S) instead of the exact-fp32 MFMA."""
    ctx = ctx or _lib.default_context()
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w_ohwi, dtype=np.float32)
    cout, k, _, wcin = w.shape
    cin = wcin if cin is None else cin
    assert cin == wcin
    n, h, wd, cs = x.shape
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    if out is None:
        out = np.zeros((n, ho, wo, cout), dtype=x.dtype)
    out = np.ascontiguousarray(out)
    assert out.shape[:3] == (n, ho, wo) and out.dtype == x.dtype
    if split and x.dtype != np.float32:
        raise TypeError("split=True needs float32 activations")
    d = ConvDesc(dtype=GTX_F32S if split else _dt(x), n=n, h=h, w=wd, cin=cin, cout=cout, ksize=k, stride=stride, act=int(act),
                 in_cstride=cs, in_coff=in_coff, out_cstride=out.shape[3], out_coff=out_coff,
                 has_residual=int(residual is not None))
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    r = None if residual is None else np.ascontiguousarray(residual, dtype=x.dtype)
    check(ctx.lib.gtx_op_conv2d(ctx.handle, C.byref(d), ptr(x), ptr(w), ptr(b), ptr(r), ptr(out)))
    return out


def conv2d_time(dtype, n, h, w, cin, cout, ksize, stride, iters=20, ctx=None):
    """Mean kernel time (ms) and algorithmic FLOPs of one conv launch on zero-filled data."""
    ctx = ctx or _lib.default_context()
    d = ConvDesc(dtype=dtype, n=n, h=h, w=w, cin=cin, cout=cout, ksize=ksize, stride=stride, act=1,
                 in_cstride=cin, in_coff=0, out_cstride=cout, out_coff=0, has_residual=0)
    ms, fl = C.c_float(), C.c_double()
    check(ctx.lib.gtx_op_conv2d_time(ctx.handle, C.byref(d), iters, C.byref(ms), C.byref(fl)))
    return ms.value, fl.value


def sppf_pool(x: np.ndarray, c: int, ctx=None, split: bool = False) -> np.ndarray:
    """x: [n,h,w,4c]; fills channels [c,4c) with the 5/9/13 window maxima of channels [0,c). split (float32 only): the
    device tensor is in the pair format of the default fp32 path (values come back as hi + lo, i.e. 22-bit rounded)."""
    ctx = ctx or _lib.default_context()
    x = np.ascontiguousarray(x).copy()
    n, h, w, cs = x.shape
    assert cs == 4 * c and (not split or x.dtype == np.float32)
    check(ctx.lib.gtx_op_sppf_pool(ctx.handle, GTX_F32S if split else _dt(x), n, h, w, c, ptr(x)))
    return x


def upsample2x(x: np.ndarray, c: int, in_coff: int, out: np.ndarray, out_coff: int, ctx=None) -> np.ndarray:
    ctx = ctx or _lib.default_context()
    x = np.ascontiguousarray(x)
    out = np.ascontiguousarray(out).copy()
    n, h, w, cs = x.shape
    check(ctx.lib.gtx_op_upsample2x(ctx.handle, _dt(x), n, h, w, c, ptr(x), cs, in_coff, ptr(out), out.shape[3], out_coff))
    return out


def preprocess(frame_bgr: np.ndarray, net_h: int, net_w: int, dtype=np.float32, want_gray: bool = True, ctx=None):
    """Letterbox + BGR->RGB + /255 into [net_h,net_w,4] (RGB0) and the half-res gray image."""
    ctx = ctx or _lib.default_context()
    frame = np.ascontiguousarray(frame_bgr, dtype=np.uint8)
    h, w, _ = frame.shape
    img = np.zeros((net_h, net_w, 4), dtype=dtype)
    gray = np.zeros((h // 2, w // 2), dtype=np.uint8) if want_gray else None
    check(ctx.lib.gtx_op_preprocess(ctx.handle, _dt(img), ptr(frame), h, w, net_h, net_w, ptr(img), ptr(gray),
                                    h // 2, w // 2))
    return img, gray


def match_2nn(query: np.ndarray, train: np.ndarray, iters: int = 0, ctx: _lib.Context | None = None):
    """2 nearest train rows (L2) of every query row; unit-norm [n,128] fp32 descriptors.
    -> (idx1, idx2, d1, d2[, ms_per_pass when iters > 0])."""
    ctx = ctx or _lib.default_context()
    q = np.ascontiguousarray(query, np.float32).reshape(-1, 128)
    t = np.ascontiguousarray(train, np.float32).reshape(-1, 128)
    nq, nt = len(q), len(t)
    i1, i2 = np.full(nq, -1, np.int32), np.full(nq, -1, np.int32)
    d1, d2 = np.zeros(nq, np.float32), np.zeros(nq, np.float32)
    ms = C.c_float()
    check(ctx.lib.gtx_op_match_2nn(ctx.handle, ptr(q), nq, ptr(t if nt else np.zeros((1, 128), np.float32)), nt, ptr(i1), ptr(i2),
                                   ptr(d1), ptr(d2), iters, C.byref(ms)))
    return (i1, i2, d1, d2, ms.value) if iters > 0 else (i1, i2, d1, d2)


def clahe(gray: np.ndarray, ctx=None) -> np.ndarray:
    """cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(gray) on the GPU (the stabilizer's `clahe: true` step)."""
    ctx = ctx or _lib.default_context()
    g = np.ascontiguousarray(gray, dtype=np.uint8)
    out = np.empty_like(g)
    check(ctx.lib.gtx_op_clahe(ctx.handle, ptr(g), g.shape[0], g.shape[1], ptr(out)))
    return out


def conv_xcd_ranges(blocks, cin):
    """How a grouped convolution launch of len(blocks) members is cut over the 8 XCDs (host only).
    -> (xcd_begin [9] int32, grid_blocks)."""
    lib = _lib.load()
    b = np.ascontiguousarray(blocks, dtype=np.int32)
    c = np.ascontiguousarray(cin, dtype=np.int32)
    out = np.zeros(9, np.int32)
    grid = C.c_int()
    check(lib.gtx_op_conv_xcd_ranges(len(b), b.ctypes.data, c.ctypes.data, out.ctypes.data, C.addressof(grid)))
    return out, grid.value
