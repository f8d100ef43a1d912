"""Projective helpers over the C ABI (host, f64): box warp of the stabilizer and the point
transform of the georeference stage.

Reference calls replaced: stabilo ``Stabilizer.transform_cur_boxes()`` (geotrax/extract.py:183)
and ``cv2.perspectiveTransform`` inside ``apply_homography`` (geotrax/georeference.py:599-605);
``ortho2geo`` (georeference.py:608-615) is plain affine arithmetic and is restated in numpy.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check, ptr


def warp_boxes(H: np.ndarray, xywh: np.ndarray) -> np.ndarray:
    """Each xywh box -> axis-aligned hull of its four corners mapped through H, as xywh float32."""
    lib = _lib.load()
    Hm = np.ascontiguousarray(H, dtype=np.float64).reshape(9)
    b = np.ascontiguousarray(xywh, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(b)
    check(lib.gtx_warp_boxes(ptr(Hm), ptr(b), len(b), ptr(out)))
    return out


def apply_homography(input_x: np.ndarray, input_y: np.ndarray, homography: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Same name and argument order as the reference function (georeference.py:599-605)."""
    lib = _lib.load()
    Hm = np.ascontiguousarray(homography, dtype=np.float64).reshape(9)
    x = np.ascontiguousarray(input_x, dtype=np.float64).reshape(-1)
    y = np.ascontiguousarray(input_y, dtype=np.float64).reshape(-1)
    ox, oy = np.empty_like(x), np.empty_like(y)
    check(lib.gtx_perspective_points(ptr(Hm), ptr(x), ptr(y), len(x), ptr(ox), ptr(oy)))
    return ox, oy


def ortho2geo(ortho_x: np.ndarray, ortho_y: np.ndarray, ortho_params: tuple) -> tuple[np.ndarray, np.ndarray]:
    """Orthophoto pixel -> (latitude, longitude) by the orthophoto's affine geotransform
    (georeference.py:608-615)."""
    lng0, lat0, dlng, dlat, skew_x, skew_y = ortho_params
    return lat0 + dlat * ortho_y + skew_y * ortho_x, lng0 + dlng * ortho_x + skew_x * ortho_y
