"""ORACLE -- test infrastructure, not product code.

numpy restatement of the colour conversion between a decoded video frame and what the reference's loop receives:
`cv2.VideoCapture.read()` (geotrax/extract.py:146) returns packed BGR; decoders produce planar YUV 4:2:0. OpenCV is
not vendored in /root/reference nor installed here; this follows the published fixed-point arithmetic of
`cv2.cvtColor(yuv, COLOR_YUV2BGR_I420)` (modules/imgproc/src/color_yuv.simd.hpp: ITU-R BT.601 limited range,
ITUR_BT_601_CY = 1220542, _CUB = 2116026, _CUG = -409993, _CVG = -852492, _CVR = 1673527, _SHIFT = 20; one chroma
sample per 2x2 luma block, no interpolation):

    y' = max(0, Y - 16) * CY
    B = sat8((y' + (1 << 19) + CUB * (U - 128)) >> 20)
    G = sat8((y' + (1 << 19) + CVG * (V - 128) + CUG * (U - 128)) >> 20)
    R = sat8((y' + (1 << 19) + CVR * (V - 128)) >> 20)

Held against skimage.color.ycbcr2rgb (float BT.601 limited range): <= 1 level (tests/test_independent.py).
PARITY UNPINNED against OpenCV / FFmpeg's swscale (what VideoCapture's FFmpeg backend really runs; its own tables
differ from cvtColor's in the last bit for some inputs). Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

CY, CUB, CUG, CVG, CVR, SHIFT = 1220542, 2116026, -409993, -852492, 1673527, 20


def i420_to_bgr(data: np.ndarray, h: int, w: int) -> np.ndarray:
    ch, cw = (h + 1) // 2, (w + 1) // 2
    d = np.asarray(data, np.uint8).ravel().astype(np.int64)
    Y = d[:h * w].reshape(h, w)
    U = d[h * w:h * w + ch * cw].reshape(ch, cw)
    V = d[h * w + ch * cw:h * w + 2 * ch * cw].reshape(ch, cw)
    yy, xx = np.mgrid[0:h, 0:w]
    u, v = U[yy >> 1, xx >> 1] - 128, V[yy >> 1, xx >> 1] - 128
    yl = np.maximum(Y - 16, 0) * CY
    half = 1 << (SHIFT - 1)
    out = np.stack([(yl + half + CUB * u) >> SHIFT, (yl + half + CVG * v + CUG * u) >> SHIFT, (yl + half + CVR * v) >> SHIFT], -1)
    return np.clip(out, 0, 255).astype(np.uint8)
