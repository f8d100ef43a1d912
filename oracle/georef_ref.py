"""ORACLE -- test infrastructure, not product code.

numpy restatement of the per-row transform chain of the georeference stage, geotrax/georeference.py:173-177:

    apply_homography (:599-605)   cv2.perspectiveTransform on (N,1,2) float64: (x', y') = (h11 x + h12 y + h13, h21 x + h22 y +
                                  h23) / w with w = h31 x + h32 y + h33, and (0, 0) where |w| <= DBL_EPSILON (OpenCV's
                                  published rule, modules/core/src/matmul.simd.hpp perspectiveTransform_)
    ortho2geo (:608-615)          the affine geotransform, exactly as written in the reference
    geo2local (:618-628)          geopandas/pyproj `to_crs`. pyproj is not installed here; the projected systems the pipeline
                                  uses are transverse Mercator, restated here from Snyder, "Map Projections -- A Working Manual"
                                  (USGS PP 1395, 1987), eqs. 3-21, 8-9 .. 8-15: the classical series in powers of the longitude
                                  difference -- deliberately NOT the Krueger n-series the product uses (csrc/georef.hip,
                                  geotrax_amd/georeference.py), so the two share no formula. Truncation error of this series:
                                  below 0.1 mm within 1 degree of the central meridian, ~1 mm at 3 degrees.
Pinned on: the reference tests' known answers (tests/test_georeference.py:31-63 of the reference, incl. pyproj's UTM 31N
value) and the golden Ortho_X/Y of data/results-full (tests/test_geometry.py). Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

A_WGS84, F_WGS84, F_GRS80 = 6378137.0, 1 / 298.257223563, 1 / 298.257222101


def apply_homography(x, y, H):
    x, y, H = np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(H, np.float64)
    w = H[2, 0] * x + H[2, 1] * y + H[2, 2]
    ok = np.abs(w) > np.finfo(np.float64).eps
    ws = np.where(ok, w, 1.0)
    return np.where(ok, (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / ws, 0.0), np.where(ok, (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / ws, 0.0)


def ortho2geo(ox, oy, ortho_params):
    lng0, lat0, dlng, dlat, skew_x, skew_y = ortho_params
    return lat0 + dlat * oy + skew_y * ox, lng0 + dlng * ox + skew_x * oy            # (latitude, longitude)


def _meridian_arc(phi, a, e2):
    e4, e6 = e2 * e2, e2 ** 3
    return a * ((1 - e2 / 4 - 3 * e4 / 64 - 5 * e6 / 256) * phi - (3 * e2 / 8 + 3 * e4 / 32 + 45 * e6 / 1024) * np.sin(2 * phi)
                + (15 * e4 / 256 + 45 * e6 / 1024) * np.sin(4 * phi) - (35 * e6 / 3072) * np.sin(6 * phi))


def transverse_mercator(lat_deg, lon_deg, lon0_deg, lat0_deg, k0, fe, fn, f, a=A_WGS84):
    """Snyder eqs. 8-9, 8-10 with 3-21 (meridional distance). -> (easting, northing)."""
    phi, lam = np.radians(np.asarray(lat_deg, np.float64)), np.radians(np.asarray(lon_deg, np.float64) - lon0_deg)
    e2 = f * (2 - f)
    ep2 = e2 / (1 - e2)
    N = a / np.sqrt(1 - e2 * np.sin(phi) ** 2)
    T, Cc, Aa = np.tan(phi) ** 2, ep2 * np.cos(phi) ** 2, lam * np.cos(phi)
    M, M0 = _meridian_arc(phi, a, e2), _meridian_arc(np.radians(lat0_deg), a, e2)
    x = k0 * N * (Aa + (1 - T + Cc) * Aa ** 3 / 6 + (5 - 18 * T + T ** 2 + 72 * Cc - 58 * ep2) * Aa ** 5 / 120)
    y = k0 * (M - M0 + N * np.tan(phi) * (Aa ** 2 / 2 + (5 - T + 9 * Cc + 4 * Cc ** 2) * Aa ** 4 / 24
                                         + (61 - 58 * T + T ** 2 + 600 * Cc - 330 * ep2) * Aa ** 6 / 720))
    return fe + x, fn + y


def geo2local(lat, lon, source_crs: str, target_crs: str):
    src, dst = (int(str(c).lower().split(":")[1]) for c in (source_crs, target_crs))
    assert src in (4326, 4737, 4019)
    if 32601 <= dst <= 32660 or 32701 <= dst <= 32760:
        return transverse_mercator(lat, lon, (dst % 100) * 6 - 183, 0.0, 0.9996, 500000.0, 0.0 if dst < 32700 else 10000000.0, F_WGS84)
    if dst in (5185, 5186, 5187, 5188):                        # Korea 2000 belts: origin 38 N, k0 = 1, FE 200 km, FN 600 km
        return transverse_mercator(lat, lon, {5185: 125.0, 5186: 127.0, 5187: 129.0, 5188: 131.0}[dst], 38.0, 1.0, 200000.0, 600000.0, F_GRS80)
    raise NotImplementedError(target_crs)
