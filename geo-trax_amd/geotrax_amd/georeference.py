"""Per-track georeferencing arithmetic of the `georeference` stage (host numpy; defines the CSV schema).

Reference behaviour replaced: geotrax/georeference.py:599-876 -- the point
transforms (`apply_homography`, `ortho2geo`, `geo2local`, `ortho2local`, `frame2local`), vehicle
dimensions in metres, in-frame visibility, speed / acceleration with Gaussian or Savitzky-Golay
smoothing, road-section / lane lookup and the formatting + rounding rules of the georeferenced CSV.
The heavy step of that stage, image registration, is `geotrax_amd.registration` (GPU); everything here
is O(rows) work on the track table and stays on the host, like in the reference.

`geo2local` in the reference goes through geopandas/pyproj (absent here). This module projects
WGS 84 / GRS 80 geographic coordinates with its own transverse-Mercator series (Karney-Krueger, 6th
order: sub-millimetre inside a zone) for the projected systems the pipeline uses: UTM (EPSG:326xx /
327xx) and Korea 2000 central belt 2010 (EPSG:5186, the reference default). Other targets raise.
"""
from __future__ import annotations

import logging
import math
from pathlib import Path

import numpy as np

from .geometry import apply_homography, ortho2geo  # noqa: F401  (K12 point transforms, pinned on the golden CSV)

_A = 6378137.0                     # WGS 84 / GRS 80 semi-major axis
_F_WGS84 = 1 / 298.257223563
_F_GRS80 = 1 / 298.257222101


# --------------------------------------------------------------------------- projections

def _tm_params(f: float):
    n = f / (2 - f)
    n2, n3, n4, n5, n6 = n * n, n ** 3, n ** 4, n ** 5, n ** 6
    A = _A / (1 + n) * (1 + n2 / 4 + n4 / 64 + n6 / 256)
    alpha = (
        n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800,
        13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360,
        61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440,
        49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600,
        34729 * n5 / 80640 - 3418889 * n6 / 1995840,
        212378941 * n6 / 319334400,
    )
    return A, alpha


def _tm_forward(lat_deg, lon_deg, lon0_deg, f):
    """Krueger series: (xi, eta) * A = (northing from the equator, easting from the central meridian) at k = 1."""
    A, alpha = _tm_params(f)
    e = math.sqrt(f * (2 - f))
    phi = np.radians(np.asarray(lat_deg, np.float64))
    lam = np.radians(np.asarray(lon_deg, np.float64) - lon0_deg)
    s = np.sin(phi)
    t = np.sinh(np.arctanh(s) - e * np.arctanh(e * s))            # tan of the conformal latitude
    xi_p = np.arctan2(t, np.cos(lam))
    eta_p = np.arcsinh(np.sin(lam) / np.hypot(t, np.cos(lam)))
    xi, eta = xi_p.copy(), eta_p.copy()
    for j, a in enumerate(alpha, start=1):
        xi += a * np.sin(2 * j * xi_p) * np.cosh(2 * j * eta_p)
        eta += a * np.cos(2 * j * xi_p) * np.sinh(2 * j * eta_p)
    return A * xi, A * eta


def _parse_epsg(crs: str) -> int:
    c = str(crs).strip().lower()
    if not c.startswith("epsg:"):
        raise NotImplementedError(f"CRS '{crs}': only EPSG codes are understood")
    return int(c.split(":", 1)[1])


def geo2local(latitude: np.ndarray, longitude: np.ndarray, source_crs: str, target_crs: str) -> tuple:
    """Geographic (deg) -> projected metres (georeference.py:618-628). source must be EPSG:4326
    (or 4737, Korea 2000 geographic: same ellipsoid within 0.1 mm)."""
    src, dst = _parse_epsg(source_crs), _parse_epsg(target_crs)
    if src not in (4326, 4737, 4019):
        raise NotImplementedError(f"source_crs EPSG:{src}: only geographic WGS 84 / GRS 80 sources are implemented")
    lat, lon = np.asarray(latitude, np.float64), np.asarray(longitude, np.float64)
    if 32601 <= dst <= 32660 or 32701 <= dst <= 32760:            # WGS 84 / UTM zone N|S
        zone = dst % 100
        north, east = _tm_forward(lat, lon, zone * 6 - 183, _F_WGS84)
        k0 = 0.9996
        return 500000.0 + k0 * east, (0.0 if dst < 32700 else 10000000.0) + k0 * north
    if dst in (5185, 5186, 5187, 5188):                            # Korea 2000 / {West, Central, East, East Sea} Belt 2010
        lon0 = {5185: 125.0, 5186: 127.0, 5187: 129.0, 5188: 131.0}[dst]
        north, east = _tm_forward(lat, lon, lon0, _F_GRS80)
        north0, _ = _tm_forward(38.0, lon0, lon0, _F_GRS80)
        return 200000.0 + east, 600000.0 + (north - north0)
    raise NotImplementedError(f"target_crs EPSG:{dst}: implemented targets are UTM (326xx/327xx) and Korea 2000 belts (5185-5188)")


def ortho2local(ortho_x, ortho_y, ortho_params, source_crs, target_crs) -> tuple:
    lat, lon = ortho2geo(ortho_x, ortho_y, ortho_params)
    return geo2local(lat, lon, source_crs, target_crs)


def frame2local(points_px: np.ndarray, homography: np.ndarray, ortho_params, source_crs, target_crs, ctx=None) -> np.ndarray:
    """Frame pixels -> local metres (georeference.py:173-177). With a `_lib.Context` the whole chain is one HIP pass
    (`transform_points`); without, the host functions above, as in the reference."""
    if ctx is not None:
        out = transform_points(points_px[:, 0], points_px[:, 1], homography, ortho_params, source_crs, target_crs, ctx=ctx)
        return np.array([out["x_local"], out["y_local"]]).T
    x, y = apply_homography(points_px[:, 0], points_px[:, 1], homography)
    xl, yl = ortho2local(x, y, ortho_params, source_crs, target_crs)
    return np.array([xl, yl]).T


def georef_chain(homography, ortho_params, source_crs: str | None = None, target_crs: str | None = None):
    """The parameter block of gtx_op_georef_points for a homography, an orthophoto geotransform and (optionally) a
    projected target CRS -- the same CRS rules as geo2local."""
    from ._lib import GeorefChain

    ch = GeorefChain()
    ch.H[:] = [float(v) for v in np.asarray(homography, np.float64).reshape(9)]
    ch.ortho[:] = [float(v) for v in ortho_params]
    ch.projected = 0
    if target_crs is None:
        return ch
    src, dst = _parse_epsg(source_crs), _parse_epsg(target_crs)
    if src not in (4326, 4737, 4019):
        raise NotImplementedError(f"source_crs EPSG:{src}: only geographic WGS 84 / GRS 80 sources are implemented")
    ch.projected, ch.semi_major = 1, _A
    if 32601 <= dst <= 32660 or 32701 <= dst <= 32760:
        ch.flattening, ch.lon0_deg, ch.k0 = _F_WGS84, (dst % 100) * 6 - 183, 0.9996
        ch.false_easting, ch.false_northing = 500000.0, 0.0 if dst < 32700 else 10000000.0
    elif dst in (5185, 5186, 5187, 5188):
        lon0 = {5185: 125.0, 5186: 127.0, 5187: 129.0, 5188: 131.0}[dst]
        north0, _ = _tm_forward(38.0, lon0, lon0, _F_GRS80)
        ch.flattening, ch.lon0_deg, ch.k0 = _F_GRS80, lon0, 1.0
        ch.false_easting, ch.false_northing = 200000.0, 600000.0 - float(north0)
    else:
        raise NotImplementedError(f"target_crs EPSG:{dst}: implemented targets are UTM (326xx/327xx) and Korea 2000 belts (5185-5188)")
    return ch


def transform_points(x_px, y_px, homography, ortho_params, source_crs: str | None = None, target_crs: str | None = None, ctx=None) -> dict:
    """The per-row chain of the georeference stage on the GPU (gtx_op_georef_points, include/gtx.h): frame pixel ->
    orthophoto pixel -> latitude/longitude -> local metres, f64, one pass over the track table.
    -> dict(ortho_x, ortho_y, latitude, longitude[, x_local, y_local])."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    ctx = ctx or _lib.default_context()
    ch = georef_chain(homography, ortho_params, source_crs, target_crs)
    x = np.ascontiguousarray(x_px, dtype=np.float64).reshape(-1)
    y = np.ascontiguousarray(y_px, dtype=np.float64).reshape(-1)
    if len(x) != len(y):
        raise ValueError("x and y differ in length")
    names = ["ortho_x", "ortho_y", "latitude", "longitude"] + (["x_local", "y_local"] if ch.projected else [])
    out = {k: np.empty_like(x) for k in names}
    ptrs = [_lib.ptr(out[k]) for k in names] + [None] * (6 - len(names))
    _lib.check(lib.gtx_op_georef_points(ctx.handle, C.byref(ch), _lib.ptr(x), _lib.ptr(y), len(x), *ptrs))
    return out


# --------------------------------------------------------------------------- per-track quantities

def convert_dimensions(track_ids, veh_dim_px, frame_size, homography, ortho_params, source_crs, target_crs) -> tuple:
    """Pixel length/width -> metres (:651-680): half-extents are laid from the frame centre along +x / +y,
    pushed through frame -> ortho -> local, and doubled. One value per track (its first row decides)."""
    length_px, width_px = np.asarray(veh_dim_px, np.float64).T
    n = len(length_px)
    length_m, width_m = np.full(n, np.nan), np.full(n, np.nan)
    ids, first, inverse = np.unique(track_ids, return_index=True, return_inverse=True)
    ok = ~np.isnan(length_px[first]) & ~np.isnan(width_px[first])
    if ok.any():
        cx, cy = frame_size[1] / 2, frame_size[0] / 2
        k = int(ok.sum())
        pts = np.empty((3 * k, 2))
        pts[0::3] = (cx, cy)
        pts[1::3] = np.column_stack((np.full(k, cx), cy + width_px[first][ok] / 2))
        pts[2::3] = np.column_stack((cx + length_px[first][ok] / 2, np.full(k, cy)))
        real = frame2local(pts, homography, ortho_params, source_crs, target_crs)
        per_l, per_w = np.full(len(ids), np.nan), np.full(len(ids), np.nan)
        per_l[ok] = 2 * np.linalg.norm(real[0::3] - real[2::3], axis=1)
        per_w[ok] = 2 * np.linalg.norm(real[0::3] - real[1::3], axis=1)
        length_m, width_m = per_l[inverse], per_w[inverse]
    return length_m, width_m


def calculate_visibility(track_ids, bbox_unstab, frame_size, visibility_margin: int = 4) -> np.ndarray:
    """True where the un-stabilized box lies inside the frame with the margin (:683-702)."""
    x, y, w, h = np.asarray(bbox_unstab, np.float64).T
    fw, fh = frame_size[1], frame_size[0]
    m = visibility_margin
    return (x - w / 2 > m) & (x + w / 2 < fw - m - 1) & (y - h / 2 > m) & (y + h / 2 < fh - m - 1)


def interpolate_missing_points(frames, x, y) -> tuple:
    """Linear fill of skipped frames (:738-766) -> (x, y with the fills in place, indices of the original samples). The reference
    appends point by point; here the same values (x[i-1] + step * ((x[i] - x[i-1]) / gap), float64) are laid out in one pass."""
    frames = np.asarray(frames)
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    if len(frames) < 2:
        return x.copy(), y.copy(), np.arange(len(frames))
    gaps = (frames[1:] - frames[:-1]).astype(np.int64)
    per = np.where(gaps > 1, gaps, 1)                           # points an interval adds: its fills, then its sample
    if per.sum() == len(gaps):                                  # no frame was skipped
        return x.copy(), y.copy(), np.arange(len(frames))
    pos = np.concatenate(([0], np.cumsum(per)))                 # where the original samples land
    seg = np.repeat(np.arange(len(per)), per)                   # interval of every point after the first
    step = np.arange(1, pos[-1] + 1) - pos[seg]                 # 1 .. per[seg]; == per[seg] at the interval's own sample
    with np.errstate(divide="ignore", invalid="ignore"):
        dx, dy = (x[1:] - x[:-1]) / gaps, (y[1:] - y[:-1]) / gaps
    fill = step < per[seg]
    xs, ys = np.empty(pos[-1] + 1), np.empty(pos[-1] + 1)
    xs[0], ys[0] = x[0], y[0]
    xs[1:] = np.where(fill, x[seg] + step * dx[seg], x[seg + 1])
    ys[1:] = np.where(fill, y[seg] + step * dy[seg], y[seg + 1])
    return xs, ys, pos


def compute_speed(x, y, fps: float) -> np.ndarray:
    return np.sqrt(np.diff(x) ** 2 + np.diff(y) ** 2) * fps


def compute_acceleration(speed, fps: float) -> np.ndarray:
    return np.diff(speed) * fps


def apply_filter(data, kernel_size: int, filter_type: str = "gaussian") -> np.ndarray:
    """Gaussian (sigma = kernel_size, reflect, truncate 3) or Savitzky-Golay (odd window, order 2, nearest) (:788-799)."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal import savgol_filter

    if filter_type == "gaussian":
        return gaussian_filter1d(data, kernel_size, mode="reflect", truncate=3.0)
    if filter_type == "savgol":
        window = kernel_size if kernel_size % 2 == 1 else kernel_size + 1
        return savgol_filter(data, window_length=window, polyorder=2, mode="nearest")
    raise ValueError(f"Invalid filter type: '{filter_type}'. Supported types: 'gaussian', 'savgol'.")


def compute_kinematics(track_ids, frame_num, x_local, y_local, visibility, fps: float, filter_type: str, kernel_size: int,
                       is_interpolated=None, conversion_factor: float = 3.6) -> tuple:
    """Speed (km/h) and acceleration (m/s^2) per row from the visible, really-detected points of each
    track (:705-735): at least 3 such points, gaps filled linearly for the differentiation only, the
    speed smoothed, first speed and first two accelerations undefined."""
    n = len(track_ids)
    speed, accel = np.full(n, np.nan), np.full(n, np.nan)
    track_ids = np.asarray(track_ids)
    order = np.argsort(track_ids, kind="stable")               # per track the ascending row indices np.where(track_ids == tid)[0] gives,
    bounds = np.flatnonzero(track_ids[order][1:] != track_ids[order][:-1]) + 1   # without a pass over the table per track
    for idx in (np.split(order, bounds) if n else []):
        real = (np.asarray(is_interpolated)[idx] == 0) if is_interpolated is not None else np.ones(len(idx), bool)
        use = np.asarray(visibility)[idx] & real
        if use.sum() < 3:
            continue
        xs, ys, present = interpolate_missing_points(frame_num[idx][use], x_local[idx][use], y_local[idx][use])
        v = apply_filter(compute_speed(xs, ys, fps), kernel_size, filter_type)
        a = compute_acceleration(v, fps)
        v = np.insert(v * conversion_factor, 0, np.nan)
        a = np.insert(a, 0, [np.nan] * 2)
        speed[idx[use]] = v[present]
        accel[idx[use]] = a[present]
    return speed, accel


def _points_in_quad(px, py, quad) -> np.ndarray:
    """Points strictly inside a simple quadrilateral (shapely 'within': boundary excluded), any vertex order."""
    q = np.asarray(quad, np.float64).reshape(4, 2)
    inside = np.zeros(len(px), bool)
    on_edge = np.zeros(len(px), bool)
    for i in range(4):
        (x1, y1), (x2, y2) = q[i], q[(i + 1) % 4]
        cross = (x2 - x1) * (py - y1) - (y2 - y1) * (px - x1)
        within_seg = (np.minimum(x1, x2) <= px) & (px <= np.maximum(x1, x2)) & (np.minimum(y1, y2) <= py) & (py <= np.maximum(y1, y2))
        on_edge |= (cross == 0) & within_seg
        straddles = (y1 > py) != (y2 > py)
        with np.errstate(divide="ignore", invalid="ignore"):
            xint = x1 + (py - y1) * (x2 - x1) / (y2 - y1)
        inside ^= straddles & (px < xint)
    return inside & ~on_edge


def assign_road_section_lane(ortho_x, ortho_y, ortho_segmentation) -> tuple:
    """Section / lane of the first segmentation polygon (file order) that contains each point (:458-479);
    NaN where none does. `ortho_segmentation`: DataFrame whose first ten columns are section, lane and the
    tl/bl/br/tr corner coordinates. (None, None) for an empty table."""
    if ortho_segmentation is None or len(ortho_segmentation) == 0:
        return None, None
    seg = ortho_segmentation.iloc[:, :10]
    px, py = np.asarray(ortho_x, np.float64), np.asarray(ortho_y, np.float64)
    section = np.full(len(px), np.nan, dtype=object)
    lane = np.full(len(px), np.nan, dtype=object)
    taken = np.zeros(len(px), bool)
    for row in seg.itertuples(index=False):
        hit = _points_in_quad(px, py, row[2:10]) & ~taken
        section[hit], lane[hit] = row[0], row[1]
        taken |= hit
    return section, lane


# --------------------------------------------------------------------------- output table

_ROUND = {"Ortho_X": 1, "Ortho_Y": 1, "Local_X": 2, "Local_Y": 2, "Longitude": 7, "Latitude": 7, "Vehicle_Length": 2,
          "Vehicle_Width": 2, "Vehicle_Speed": 1, "Vehicle_Acceleration": 2}


def create_and_format_georeferenced_df(track_id, timestamps, frame_num, x_stab_ortho, y_stab_ortho, x_local, y_local, latitude,
                                       longitude, veh_dim_real, class_id, v_speed, v_acceleration, road_section, lane_number,
                                       visibility, min_traj_length, is_interpolated=None, *, logger: logging.Logger):
    """Column set, order, rounding and the minimum-trajectory filter of the georeferenced CSV (:802-866)."""
    import pandas as pd

    cols = {
        "Vehicle_ID": track_id, "Timestamp": timestamps if np.size(timestamps) > 0 else None, "Frame_Number": frame_num,
        "Ortho_X": x_stab_ortho, "Ortho_Y": y_stab_ortho, "Local_X": x_local, "Local_Y": y_local, "Latitude": latitude,
        "Longitude": longitude, "Vehicle_Length": veh_dim_real[0], "Vehicle_Width": veh_dim_real[1], "Vehicle_Class": class_id,
        "Vehicle_Speed": v_speed, "Vehicle_Acceleration": v_acceleration, "Road_Section": road_section, "Lane_Number": lane_number,
        "Visibility": visibility, "Is_Interpolated": is_interpolated,
    }
    # copy=False: the columns stay the arrays they are (no consolidation into per-dtype blocks: 2.7 s of the 4.9 s this function took
    # on a 900 k-row video); every column below is replaced by a new array, never written in place
    df = pd.DataFrame({k: v for k, v in cols.items() if v is not None}, copy=False)
    for name, digits in _ROUND.items():
        df[name] = np.round(df[name], digits)
    df["Visibility"] = df["Visibility"].astype("int")
    if "Is_Interpolated" in df.columns:
        df["Is_Interpolated"] = df["Is_Interpolated"].astype("int")
    if "Lane_Number" in df.columns:
        # str(int(v)) for a lane number, "" for a missing one (:845-846) -- through the few distinct values instead of row by row
        lane = df["Lane_Number"]
        if lane.dtype.kind in "fiu":
            v = lane.to_numpy()
            have = ~np.isnan(v) if v.dtype.kind == "f" else np.ones(len(v), bool)
            uniq, inv = np.unique(v[have].astype(np.int64), return_inverse=True)
            out = np.full(len(v), "", dtype=object)
            out[have] = np.array([str(int(u)) for u in uniq], dtype=object)[inv] if len(uniq) else []
            df["Lane_Number"] = out
        else:
            df["Lane_Number"] = lane.apply(lambda v: str(int(v)) if pd.notna(v) else "")
    if min_traj_length > 0:
        before = df["Vehicle_ID"].nunique()
        if "Is_Interpolated" in df.columns:
            real = (df["Is_Interpolated"] == 0).groupby(df["Vehicle_ID"]).transform("sum")
            df = df[real >= min_traj_length]
        else:
            df = df[df.groupby("Vehicle_ID")["Vehicle_ID"].transform("size") >= min_traj_length]
        removed = before - df["Vehicle_ID"].nunique()
        if removed > 0:
            logger.info(f"Removed {removed} vehicles with fewer than {min_traj_length} detected points.")
    logger.info("Georeferenced DataFrame successfully created and formatted.")
    return df


def save_georeferenced_data(path: Path, georeferenced_df, logger: logging.Logger) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    from . import tables

    if not tables.dataframe_to_csv(path, georeferenced_df):     # int64 / float64 / label columns: the library's writer, same bytes
        georeferenced_df.to_csv(path, index=False)               # anything else in the frame (free text, dates): pandas' own
    logger.info(f"Georeferenced data saved to: '{path}'.")


def save_homography(path: Path, homography: np.ndarray, logger: logging.Logger) -> None:
    """The nine entries on one comma-separated line, 20 significant digits (georeference.py:879-889)."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    from . import tables

    tables.savetxt(path, np.asarray(homography, np.float64).reshape(1, -1), 20)     # np.savetxt(..., fmt="%.20g", delimiter=",")
    logger.info(f"Reference-to-orthophoto homography saved to: '{path}'.")
