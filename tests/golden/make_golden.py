#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the reference at /root/reference.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Two kinds of fixture, both *data* (inputs + expected outputs), never reference source:

1. `postprocess_vectors.npz` -- seeded inputs pushed through the reference's own pure-numpy
   post-processing functions (geotrax/extract.py:273-484), imported here with stub modules for the
   third-party packages that are not installed (cv2, stabilo, ultralytics, tqdm). These pin the
   build's restatement of aggregate_results / remove_short_tracks / calculate_unique_classes /
   estimate_vehicle_dimensions / interpolate_tracks / postprocess_tracks to the real code.
2. The reference's committed golden outputs for data/U_video_cut.mp4 (the only result-level pins
   of the detect/track/stabilize chain), stored compactly:
   `U_video_cut.txt.gz`, `U_video_cut_vid_transf.txt`, `U_video_cut_geo_transf.txt`,
   `U_video_cut_csv_cols.npz` (the CSV columns the georeference parity check uses).
3. `georeference_vectors.npz` + `georeference_table.csv` -- seeded track tables pushed through the
   reference's own per-track georeference functions (geotrax/georeference.py:683-866: visibility,
   kinematics with both filters, gap interpolation, table formatting / rounding / minimum-trajectory
   filter), imported with stubs for cv2 / geopandas / shapely / tqdm. CRS reprojection and lane lookup need
   pyproj / shapely themselves and are pinned on the reference tests' known answers instead.
"""
import argparse
import gzip
import logging
import shutil
import sys
import types
from pathlib import Path
from unittest.mock import patch

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


def import_reference_extract():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("cv2", VideoCapture=object, CAP_PROP_FRAME_COUNT=7, CAP_PROP_FRAME_WIDTH=3, CAP_PROP_FRAME_HEIGHT=4)
    stub("stabilo", Stabilizer=object)
    stub("ultralytics", YOLO=object, RTDETR=object)
    stub("ultralytics.utils")
    stub("ultralytics.utils.checks", check_yolo=lambda **k: None)
    stub("ultralytics.utils.files", increment_path=lambda p, **k: p)
    try:
        import tqdm  # noqa: F401
    except ImportError:
        stub("tqdm", tqdm=object)
    try:
        import huggingface_hub  # noqa: F401
    except ImportError:
        stub("huggingface_hub", hf_hub_download=lambda **k: None)
    sys.path.insert(0, str(REF))
    import geotrax.extract as ex

    return ex


def random_tracks(rng, n_tracks, n_frames, w_img=3840, h_img=2160, with_gaps=False):
    """A plausible [N,12] float32 track table: frame,id,xywh,xywh_stab,cls,conf."""
    rows = []
    for tid in range(1, n_tracks + 1):
        t0 = int(rng.integers(0, max(n_frames // 2, 1)))
        length = int(rng.integers(1, n_frames - t0 + 1))
        x, y = rng.uniform(0, w_img), rng.uniform(0, h_img)
        horizontal = rng.random() < 0.6
        w, h = (rng.uniform(60, 160), rng.uniform(28, 60)) if horizontal else (rng.uniform(28, 60), rng.uniform(60, 160))
        speed = rng.choice([0.0, rng.uniform(0.5, 9.0)])
        ang = (0 if horizontal else np.pi / 2) + rng.normal(0, 0.08) + (np.pi if rng.random() < 0.5 else 0)
        if rng.random() < 0.15:
            ang = rng.uniform(0, 2 * np.pi)  # diagonal mover: fails the azimuth gate
        base_cls = int(rng.integers(0, 4))
        for k in range(length):
            if with_gaps and 0 < k < length - 1 and rng.random() < 0.12:
                continue
            f = t0 + k
            cx, cy = x + speed * k * np.cos(ang), y - speed * k * np.sin(ang)
            bw, bh = w * (1 + rng.normal(0, 0.03)), h * (1 + rng.normal(0, 0.03))
            sx, sy = cx + 0.02 * f, cy + 0.04 * f
            cls = base_cls if rng.random() < 0.85 else int(rng.integers(0, 4))
            rows.append([f, tid, cx, cy, bw, bh, sx, sy, bw * 1.001, bh * 1.001, cls, rng.uniform(0.26, 0.95)])
    a = np.asarray(rows, dtype=np.float32)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def make_postprocess_vectors(ex):
    logger = logging.getLogger("golden")
    rng = np.random.default_rng(20240607)
    out = {}
    dim_cfgs = [
        dict(eps=4, r0=1.25, gsd=0.02725, theta_bar=15, tau_c={0: 1.83, 1: 2.85, 2: 1.70, 3: 1.80, -1: 1.70}),  # default.yaml:87-97
        dict(eps=5, r0=3.0, gsd=0.02725, theta_bar=15, tau_c={-1: 1.0}),                                       # tests/test_extract.py:83-94
        dict(eps=0, r0=0.6, gsd=0.05, theta_bar=30, tau_c={0: 1.2, -1: 2.0}),
    ]
    cases = []
    for ci, (n_tracks, n_frames, gaps, min_len) in enumerate([(6, 12, False, 3), (25, 40, True, 3), (60, 90, True, 5), (1, 3, False, 3)]):
        tracks = random_tracks(rng, n_tracks, n_frames, with_gaps=gaps)
        for di, dc in enumerate(dim_cfgs):
            key = f"case{ci}_dim{di}"
            cfg_main = {
                'args': argparse.Namespace(source=Path('dummy.mp4'), interpolate=True),
                'extraction': {'min_track_length': min_len, 'interpolate': True, 'dimension_estimation': dc},
                'tracker': {'active': 'botsort', 'botsort': {'track_buffer': 4}},
            }
            with patch('geotrax.extract.get_video_dimensions', return_value=(3840, 2160)):
                r_short = ex.remove_short_tracks(tracks.copy(), logger, min_len)
                r_cls = ex.calculate_unique_classes(r_short.copy())
                r_dim = ex.estimate_vehicle_dimensions(r_cls.copy(), cfg_main)
                r_int = ex.interpolate_tracks(r_dim.copy(), logger, 4)
                r_all = ex.postprocess_tracks(tracks.copy(), {'main': cfg_main}, logger)
            out[key + "_in"] = tracks
            out[key + "_short"] = r_short
            out[key + "_cls"] = r_cls
            out[key + "_dim"] = r_dim
            out[key + "_interp"] = r_int
            out[key + "_all"] = r_all
            cases.append((key, min_len, di))
    out["dim_cfg_index"] = np.asarray([c[2] for c in cases])
    out["min_len"] = np.asarray([c[1] for c in cases])
    out["case_keys"] = np.asarray([c[0] for c in cases])
    # aggregate_results: per-frame lists incl. an untracked (-1) detection and an empty frame
    fa = [np.full((3, 1), 0, dtype=np.uint32), np.full((2, 1), 2, dtype=np.uint32)]
    ti = [np.array([[1], [2], [65535]], dtype=np.uint16), np.full((2, 1), -1)]
    bb = [rng.uniform(0, 2000, (3, 4)).astype(np.float32), rng.uniform(0, 2000, (2, 4)).astype(np.float32)]
    bs = [b + np.float32(0.5) for b in bb]
    ci_ = [np.array([[0], [3], [1]], dtype=np.uint8), np.array([[2], [2]], dtype=np.uint8)]
    cf = [rng.uniform(0.25, 1, (3, 1)).astype(np.float32), rng.uniform(0.25, 1, (2, 1)).astype(np.float32)]
    tr = [np.hstack((np.array([[2]]), rng.normal(0, 1, (1, 9))))]
    tracks, transf = ex.aggregate_results([a.copy() for a in fa], [a.copy() for a in ti], [a.copy() for a in bb],
                                          [a.copy() for a in bs], [a.copy() for a in ci_], [a.copy() for a in cf],
                                          [a.copy() for a in tr], logger)
    for name, lst in (("frame", fa), ("id", ti), ("bbox", bb), ("bbox_stab", bs), ("cls", ci_), ("conf", cf)):
        for k, a in enumerate(lst):
            out[f"agg_{name}_{k}"] = a
    out["agg_transform_0"] = tr[0]
    out["agg_tracks"] = tracks
    out["agg_transforms"] = transf
    e_tracks, e_transf = ex.aggregate_results([], [], [], [], [], [], [], logger)
    out["agg_empty_tracks_shape"] = np.asarray(e_tracks.shape)
    out["agg_empty_transforms_shape"] = np.asarray(e_transf.shape)
    np.savez_compressed(OUT / "postprocess_vectors.npz", **out)
    print("postprocess_vectors.npz:", len(out), "arrays")


def make_class_vote_ties(ex):
    """class_vote_ties.npz: track tables on which the winner of the confidence-weighted class vote depends on HOW the
    scores are summed (row by row in float32, as the reference does on its float32 table, vs. any other order or
    precision): two classes per track whose vote totals differ by less than a float32 ulp of the sum. Found by seeded
    search; inputs plus the reference's own output."""
    rng = np.random.default_rng(7)
    found_in, found_out = [], []
    while len(found_in) < 12:
        n = int(rng.integers(40, 400))
        conf = rng.uniform(0.25, 0.95, n).astype(np.float32)
        cls = rng.integers(0, 2, n)
        # nudge one score so that the exact (float64) totals of the two classes almost coincide
        d = conf[cls == 0].astype(np.float64).sum() - conf[cls == 1].astype(np.float64).sum()
        k = int(np.nonzero(cls == (0 if d > 0 else 1))[0][-1])
        c = float(conf[k]) - abs(d) + rng.uniform(-2e-6, 2e-6)
        if not 0.05 < c < 0.99:
            continue
        conf[k] = np.float32(c)
        t = np.zeros((n, 12), np.float32)
        t[:, 0], t[:, 1], t[:, -2], t[:, -1] = np.arange(n), 1, cls, conf
        seq32 = [np.float32(0), np.float32(0)]
        for c_, s_ in zip(cls, conf):
            seq32[c_] = np.float32(seq32[c_] + s_)
        exact = [conf[cls == 0].astype(np.float64).sum(), conf[cls == 1].astype(np.float64).sum()]
        if int(np.argmax(seq32)) == int(np.argmax(exact)) and seq32[0] != seq32[1]:
            continue                                  # keep only tables where the summation order decides (or ties exactly)
        found_in.append(t)
        found_out.append(ex.calculate_unique_classes(t.copy())[:, -2].copy())
    np.savez_compressed(OUT / "class_vote_ties.npz", **{f"in{i}": a for i, a in enumerate(found_in)},
                        **{f"out{i}": a for i, a in enumerate(found_out)})
    print("class_vote_ties.npz:", len(found_in), "tables")


def copy_reference_goldens():
    src = REF / "data" / "results-full"
    with open(src / "U_video_cut.txt", "rb") as f, gzip.GzipFile(OUT / "U_video_cut.txt.gz", "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    shutil.copyfile(src / "U_video_cut_vid_transf.txt", OUT / "U_video_cut_vid_transf.txt")
    shutil.copyfile(src / "U_video_cut_geo_transf.txt", OUT / "U_video_cut_geo_transf.txt")
    import pandas as pd

    df = pd.read_csv(src / "U_video_cut.csv")
    np.savez_compressed(OUT / "U_video_cut_csv_cols.npz",
                        vehicle_id=df["Vehicle_ID"].to_numpy(np.int32), frame=df["Frame_Number"].to_numpy(np.int32),
                        ortho_x=df["Ortho_X"].to_numpy(np.float64), ortho_y=df["Ortho_Y"].to_numpy(np.float64),
                        vehicle_length=df["Vehicle_Length"].to_numpy(np.float64), vehicle_width=df["Vehicle_Width"].to_numpy(np.float64),
                        vehicle_class=df["Vehicle_Class"].to_numpy(np.int32))
    print("copied reference golden outputs")


def import_reference_georeference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    import_reference_extract()                       # installs the cv2 / ultralytics / stabilo stubs
    sys.modules["cv2"].USAC_MAGSAC = 38              # default argument of registration.estimate_homography
    stub("geopandas", GeoDataFrame=object, points_from_xy=lambda *a, **k: None, sjoin=lambda *a, **k: None)
    stub("shapely")
    stub("shapely.geometry", Polygon=object, Point=object)
    try:
        import PIL  # noqa: F401
    except ImportError:
        stub("PIL", Image=object, TiffImagePlugin=object)
    import geotrax.georeference as gr

    return gr


def make_georeference_vectors(gr):
    rng = np.random.default_rng(20260701)
    logger = logging.getLogger("golden")
    out = {}
    # a track table with gaps, some rows outside the frame, some interpolated rows
    tracks = random_tracks(rng, 40, 60, with_gaps=True)
    tid, frame = tracks[:, 1].astype(np.int64), tracks[:, 0].astype(np.int64)
    bbox = tracks[:, 2:6].astype(np.float64)
    bbox[rng.random(len(bbox)) < 0.08, 0] = rng.uniform(0, 30)          # near the left edge
    out["track_id"], out["frame"], out["bbox"] = tid, frame, bbox
    vis = gr.calculate_visibility(tid, bbox, (2160, 3840), 4)
    out["visibility"] = vis
    out["visibility_m10"] = gr.calculate_visibility(tid, bbox, (2160, 3840), 10)
    x_local = 200000 + 0.0268 * tracks[:, 6].astype(np.float64) + rng.normal(0, 0.01, len(tid))
    y_local = 540000 - 0.0268 * tracks[:, 7].astype(np.float64) + rng.normal(0, 0.01, len(tid))
    out["x_local"], out["y_local"] = x_local, y_local
    is_interp = (rng.random(len(tid)) < 0.1).astype(np.int64)
    out["is_interpolated"] = is_interp
    for name, ft, ks, interp in (("gauss14", "gaussian", 14, None), ("gauss3_interp", "gaussian", 3, is_interp),
                                 ("savgol7", "savgol", 7, None), ("savgol8_interp", "savgol", 8, is_interp)):
        v, a = gr.compute_kinematics(tid, frame, x_local, y_local, vis, 29.97, ft, ks, is_interpolated=interp)
        out[f"speed_{name}"], out[f"accel_{name}"] = v, a
    fr = np.array([3, 4, 7, 8, 12])
    xi, yi, present = gr.interpolate_missing_points(fr, np.array([0.0, 1.0, 5.5, 6.0, 9.0]), np.array([2.0, 2.5, 1.0, 0.0, -4.0]))
    out["interp_x"], out["interp_y"], out["interp_present"] = np.asarray(xi), np.asarray(yi), present
    # the formatted table (rounding rules, column order, min_traj_length filter on really-detected rows)
    v, a = out["speed_gauss3_interp"], out["accel_gauss3_interp"]
    lat, lon = 37.38 + 1e-6 * tracks[:, 7].astype(np.float64), 126.66 + 1e-6 * tracks[:, 6].astype(np.float64)
    lane = np.where(rng.random(len(tid)) < 0.3, np.nan, rng.integers(1, 5, len(tid)).astype(np.float64))
    section = np.where(np.isnan(lane), None, "A").astype(object)
    out["lat"], out["lon"], out["lane"] = lat, lon, lane
    out["veh_len"], out["veh_wid"] = rng.uniform(3.5, 12, len(tid)), rng.uniform(1.5, 2.6, len(tid))
    df = gr.create_and_format_georeferenced_df(tid, np.array([]), frame, tracks[:, 6].astype(np.float64) * 3.7123, tracks[:, 7].astype(np.float64) * 3.7123,
                                               x_local, y_local, lat, lon, (out["veh_len"], out["veh_wid"]), tracks[:, 10].astype(np.int64),
                                               v, a, section, lane, vis, 15, is_interpolated=is_interp, logger=logger)
    df.to_csv(OUT / "georeference_table.csv", index=False)
    out["stab_x"], out["stab_y"], out["cls"] = tracks[:, 6].astype(np.float64), tracks[:, 7].astype(np.float64), tracks[:, 10].astype(np.int64)
    np.savez_compressed(OUT / "georeference_vectors.npz", **out)
    print("georeference_vectors.npz:", len(out), "arrays;", len(df), "table rows")


if __name__ == "__main__":
    ex = import_reference_extract()
    if "--ties-only" in sys.argv:
        make_class_vote_ties(ex)
        sys.exit(0)
    make_postprocess_vectors(ex)
    make_class_vote_ties(ex)
    copy_reference_goldens()
    make_georeference_vectors(import_reference_georeference())
