"""ORACLE -- test infrastructure, not product code.

numpy restatement of `cv2.warpPerspective(frame, M, (w, h))` with the default flags (INTER_LINEAR,
BORDER_CONSTANT, value 0), the call the reference makes in its visualisation modes 1 and 4
(geotrax/visualize.py:285-289).

OpenCV (opencv-python, pulled in transitively by ultralytics/stabilo, pyproject.toml:55-66) is not vendored
in /root/reference and not installed here, so this follows its published algorithm (modules/imgproc/src/
imgwarp.cpp: `warpPerspective` -> `WarpPerspectiveInvoker` -> `remap` with INTER_BITS = 5):
  * M is inverted (no WARP_INVERSE_MAP), all coordinate arithmetic in float64;
  * the destination is walked in blocks 64 pixels wide; per row of a block X0 = M0*bx + M1*y + M2 (same for
    Y0, W0), per pixel W = W0 + M6*x1, W = 32/W (0 when W == 0), X = saturate_cast<int>((X0 + M0*x1) * W),
    i.e. round-half-to-even of the source coordinate in 1/32-pixel units, clamped to the int range;
  * the integer part selects the 2x2 neighbourhood, the 5-bit fractions the bilinear weights; the weight table
    holds the products (1-fx)(1-fy) ... scaled to 15 bits, which for 5-bit fractions are the exact integers
    32*(32-ax)(32-ay) ...; result = (sum + 2^14) >> 15 == (sum/32 + 512) >> 10; taps outside the image are 0.
PARITY UNPINNED against OpenCV itself (it cannot be imported here, and the reference's tests never call the
warp). Held against skimage.transform.warp(order=1) (exact float coordinates): <= 1 grey level except where the 1/32-pixel
coordinate quantisation meets the crop's sharpest edges (<= 2, < 0.1 % of pixels), tests/test_independent.py. One known freedom: OpenCV inverts M with its own LU routine; this file uses numpy.linalg.inv, the
library its adjugate formula -- the inverses agree to ~1e-16 relative, which can move a coordinate across a
1/32-pixel rounding boundary for about one pixel in 10^4..10^5 frames' worth.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np


def warp_perspective(src: np.ndarray, H: np.ndarray, M_inv: np.ndarray | None = None) -> np.ndarray:
    """src: [h, w, c] uint8; H: 3x3 mapping source -> destination pixels (what cv2.warpPerspective is given).
    M_inv overrides the inverse (to check a kernel against the very same matrix)."""
    h, w = src.shape[:2]
    M = (np.linalg.inv(np.asarray(H, dtype=np.float64)) if M_inv is None else np.asarray(M_inv, dtype=np.float64)).ravel()
    x = np.arange(w)
    bx, x1 = (x & ~63).astype(np.float64), (x & 63).astype(np.float64)
    y = np.arange(h, dtype=np.float64)[:, None]
    X0 = M[0] * bx[None, :] + M[1] * y + M[2]
    Y0 = M[3] * bx[None, :] + M[4] * y + M[5]
    W0 = M[6] * bx[None, :] + M[7] * y + M[8]
    W = W0 + M[6] * x1[None, :]
    with np.errstate(divide="ignore"):
        W = np.where(W != 0.0, 32.0 / W, 0.0)
    lim = lambda v: np.maximum(-2147483648.0, np.minimum(2147483647.0, v))
    X = np.rint(lim((X0 + M[0] * x1[None, :]) * W)).astype(np.int64)
    Y = np.rint(lim((Y0 + M[3] * x1[None, :]) * W)).astype(np.int64)
    x0, y0, ax, ay = X >> 5, Y >> 5, X & 31, Y & 31
    s = src.astype(np.int64)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = s[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return v * ok[..., None]

    acc = (tap(y0, x0) * ((32 - ax) * (32 - ay))[..., None] + tap(y0, x0 + 1) * (ax * (32 - ay))[..., None]
           + tap(y0 + 1, x0) * ((32 - ax) * ay)[..., None] + tap(y0 + 1, x0 + 1) * (ax * ay)[..., None])
    return ((acc + 512) >> 10).astype(np.uint8)
