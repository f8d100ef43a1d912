"""ORACLE -- test infrastructure, not product code.

numpy restatement of cv2.createCLAHE(clipLimit=2.0, tileGridSize=(8, 8)).apply(gray) on a u8 image: what stabilo's
`clahe: true` runs on the (downsampled) gray frame before feature detection (reference config `stabilo.clahe`,
geotrax/cfg/default.yaml:105; the `stable` preset switches it on). stabilo and OpenCV are not vendored in /root/reference
and cv2 is not installed here; this follows OpenCV's published algorithm (modules/imgproc/src/clahe.cpp) from memory:
  * the image is extended to a multiple of the tile grid with BORDER_REFLECT_101 for the histograms only;
  * per tile: 256-bin histogram, clipped at max(1, int(clipLimit * tileArea / 256)), the clipped mass redistributed
    (an equal batch to every bin, the residual one count every max(256 / residual, 1)-th bin from bin 0),
    lut[i] = saturate(round_half_even(cumsum[i] * float32(255 / tileArea)));
  * per pixel: the four surrounding tile LUTs blended bilinearly in float32 (tile coordinate x / tileWidth - 0.5,
    neighbours clamped at the border), round half to even, saturate.
PARITY UNPINNED against cv2 itself. Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


def _reflect101(i, n):
    i = np.abs(i)
    return np.where(i >= n, 2 * n - 2 - i, i)


def clahe_luts(gray: np.ndarray, clip_limit: float = 2.0, grid: int = 8):
    h, w = gray.shape
    if w % grid == 0 and h % grid == 0:
        ext = gray
    else:
        ys = _reflect101(np.arange(h + (grid - h % grid)), h)
        xs = _reflect101(np.arange(w + (grid - w % grid)), w)
        ext = gray[np.ix_(ys, xs)]
    th, tw = ext.shape[0] // grid, ext.shape[1] // grid
    area = th * tw
    lut_scale = np.float32(255.0) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((grid, grid, 256), np.uint8)
    for ty in range(grid):
        for tx in range(grid):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256
                residual = clipped - batch * 256
                hist += batch
                if residual:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1
                        i += step
                        residual -= 1
            cs = np.cumsum(hist).astype(np.float32) * lut_scale
            luts[ty, tx] = np.clip(np.rint(cs), 0, 255).astype(np.uint8)
    return luts, (th, tw)


def clahe(gray: np.ndarray, clip_limit: float = 2.0, grid: int = 8) -> np.ndarray:
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    luts, (th, tw) = clahe_luts(gray, clip_limit, grid)
    f32 = np.float32
    txf = np.arange(w, dtype=f32) * (f32(1.0) / f32(tw)) - f32(0.5)
    tyf = np.arange(h, dtype=f32) * (f32(1.0) / f32(th)) - f32(0.5)
    tx1 = np.floor(txf).astype(np.int32)
    ty1 = np.floor(tyf).astype(np.int32)
    xa = (txf - tx1.astype(f32)).astype(f32)
    ya = (tyf - ty1.astype(f32)).astype(f32)
    xa1, ya1 = f32(1.0) - xa, f32(1.0) - ya
    tx2, ty2 = np.minimum(tx1 + 1, grid - 1), np.minimum(ty1 + 1, grid - 1)
    tx1, ty1 = np.maximum(tx1, 0), np.maximum(ty1, 0)
    v = gray.astype(np.int64)
    l11 = luts[ty1[:, None], tx1[None, :], v].astype(f32)
    l12 = luts[ty1[:, None], tx2[None, :], v].astype(f32)
    l21 = luts[ty2[:, None], tx1[None, :], v].astype(f32)
    l22 = luts[ty2[:, None], tx2[None, :], v].astype(f32)
    top = (l11 * xa1[None, :]).astype(f32) + (l12 * xa[None, :]).astype(f32)
    bot = (l21 * xa1[None, :]).astype(f32) + (l22 * xa[None, :]).astype(f32)
    res = (top * ya1[:, None]).astype(f32) + (bot * ya[:, None]).astype(f32)
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)
