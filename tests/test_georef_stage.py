"""The georeference stage driver (geotrax_amd.georef_stage; reference geotrax/georeference.py:109-566 and
utils/file_utils.py): file conventions, parameter sources, timestamps, the master -> orthophoto cache with its MD5
guard, and the order of the chain. The two GPU steps (image registration, row transform) are replaced by recorders
here, like the reference's own tests mock its third-party calls; tests/test_georef_stage_gpu.py runs the real chain."""
import argparse
import logging
from pathlib import Path

import numpy as np
import pytest

logger = logging.getLogger("georef-stage-test")


def _args(source, **over):
    a = argparse.Namespace(source=Path(source), cfg=None, output_folder=None, log_path=None, verbose=False, ortho_folder=None, geo_source=None,
                           ref_frame=None, no_master=None, master_folder=None, recompute=None, segmentation_folder=None)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_location_id_paths_and_delimiter(tmp_path):
    from geotrax_amd import georef_stage as gs

    # file_utils.determine_location_id docstring examples
    assert gs.determine_location_id(Path("A1.mp4")) == "A"
    assert gs.determine_location_id(Path("2025-01-01_A_PM1.mp4")) == "A"
    assert gs.determine_location_id(Path("A1_AV.csv")) == "A"
    assert gs.determine_location_id(Path("U_video_cut.mp4")) == "U"
    with pytest.raises(SystemExit):
        gs.determine_location_id(Path("123.mp4"))
    src = tmp_path / "DATASET" / "day1" / "U_clip.npy"
    assert gs.build_result_path(src, "processed") == src.parent / "results" / "U_clip.txt"
    assert gs.build_result_path(src, "georeferenced", {"folder": "out", "georeferenced_postfix": "_geo"}) == src.parent / "out" / "U_clip_geo.csv"
    assert gs.build_result_path(src, "geo_transformations", {"folder": str(tmp_path / "abs")}) == tmp_path / "abs" / "U_clip_geo_transf.txt"
    (tmp_path / "ORTHOPHOTOS").mkdir()
    src.parent.mkdir(parents=True)
    assert gs.get_ortho_folder(src, None, logger) == tmp_path / "ORTHOPHOTOS"          # beside the DATASET ancestor
    assert gs.get_ortho_folder(src, tmp_path / "ORTHOPHOTOS", logger) == tmp_path / "ORTHOPHOTOS"
    with pytest.raises(SystemExit):
        gs.get_ortho_folder(tmp_path / "elsewhere" / "U.npy", None, logger)
    f = tmp_path / "t.txt"
    f.write_text("1 2 3\n4 5 6\n")
    assert gs.detect_delimiter(f) == " "
    f.write_text("1,2,3\n")
    assert gs.detect_delimiter(f) == ","


def test_ortho_parameter_sources(tmp_path):
    from geotrax_amd import georef_stage as gs

    of = tmp_path
    (of / "U.txt").write_text("# lng0 lat0 dlng dlat\n126.6 37.4 2e-7 -1.5e-7\n")
    np.save(of / "U.npy", np.zeros((40, 60, 3), np.uint8))
    assert gs.get_geo_params_source(None, of, "U", logger) == "text-file"
    assert gs.get_ortho_parameters(of, "U", "text-file", 15000, logger) == (126.6, 37.4, 2e-7, -1.5e-7, 0.0, 0.0)
    (of / "V_center.txt").write_text("7000 8000\n")
    (of / "ortho_parameters.txt").write_text("126.0 38.0 1e-7 -1e-7 1e-9 2e-9\n")
    np.save(of / "V.npy", np.zeros((30, 50, 3), np.uint8))
    assert gs.get_geo_params_source(None, of, "V", logger) == "center-text-file"
    lng0, lat0, dlng, dlat, sx, sy = gs.get_ortho_parameters(of, "V", "center-text-file", 100, logger)
    half, s = 50, 100 / 50                                       # cut-out of 100 px shown at 50 px: resolution scales by 2
    assert lng0 == pytest.approx(126.0 + (7000 - half) * 1e-7 + (8000 - half) * 1e-9)
    assert lat0 == pytest.approx(38.0 + (8000 - half) * -1e-7 + (7000 - half) * 2e-9)
    assert (dlng, dlat, sx, sy) == pytest.approx((1e-7 * s, -1e-7 * s, 1e-9 * s, 2e-9 * s))
    with pytest.raises(SystemExit):
        gs.get_geo_params_source("bogus", of, "U", logger)
    with pytest.raises(SystemExit):
        gs.get_geo_params_source(None, of, "W", logger)          # nothing there


def _write_clip(root: Path, n_frames=30, n_tracks=6):
    """A DATASET/ORTHOPHOTOS tree with a tiny clip, its tracks file, flight log, orthophoto parameters and lanes."""
    src = root / "DATASET" / "U_clip.npy"
    src.parent.mkdir(parents=True)
    np.save(src, np.full((3, 120, 200, 3), 90, np.uint8))
    rows = []
    for tid in range(1, n_tracks + 1):
        for f in range(n_frames):
            x, y = 20 + 5 * tid + 1.5 * f, 30 + 10 * tid
            rows.append([f, tid, x, y, 12, 6, x + 0.1, y - 0.1, 12, 6, tid % 4, 0.8, 11.0, 5.0])
    (src.parent / "results").mkdir()
    np.savetxt(src.parent / "results" / "U_clip.txt", np.asarray(rows), fmt="%g", delimiter=",")
    src.with_suffix(".csv").write_text("frame,timestamp\n" + "".join(f"{f},2022-10-04 10:00:{f:02d}.000\n" for f in range(n_frames - 3)))
    of = root / "ORTHOPHOTOS"
    (of / "master_frames").mkdir(parents=True)
    (of / "segmentations").mkdir()
    np.save(of / "U.npy", np.full((300, 300, 3), 80, np.uint8))
    np.save(of / "master_frames" / "U.npy", np.full((120, 200, 3), 85, np.uint8))
    (of / "U.txt").write_text("126.6412 37.3951 2.4e-7 -1.9e-7\n")
    (of / "segmentations" / "U.csv").write_text("section,lane,tlx,tly,blx,bly,brx,bry,trx,try\nA,1,0,0,0,150,400,150,400,0\nA,2,0,150,0,400,400,400,400,150\n")
    return src


def test_stage_chain_with_recorded_gpu_steps(tmp_path, monkeypatch):
    import pandas as pd

    from geotrax_amd import georef_stage as gs
    from geotrax_amd import georeference as G

    src = _write_clip(tmp_path)
    calls = []
    H_rm = np.array([[1.0, 0, 2], [0, 1.0, 3], [0, 0, 1]])
    H_mo = np.array([[1.5, 0, 10], [0, 1.5, 20], [0, 0, 1]])

    def fake_estimate(img_src, img_dst, logger, ctx=None, **kw):
        calls.append((img_src.shape, img_dst.shape, kw["max_features"], kw["filter_ratio"]))
        return (H_rm if img_dst.shape[0] == 120 else H_mo), 400, 900, (5000, 6000)

    def host_chain(x, y, H, ortho, src_crs, dst_crs, ctx=None):          # the documented host equivalents of gtx_op_georef_points
        ox, oy = G.apply_homography(np.asarray(x), np.asarray(y), H)
        lat, lon = G.ortho2geo(ox, oy, ortho)
        xl, yl = G.geo2local(lat, lon, src_crs, dst_crs)
        return dict(ortho_x=ox, ortho_y=oy, latitude=lat, longitude=lon, x_local=xl, y_local=yl)

    monkeypatch.setattr(gs, "estimate_homography", fake_estimate)
    monkeypatch.setattr(G, "transform_points", host_chain)
    gs.georeference(_args(src), logger)
    # reference -> master, then master -> orthophoto (computed and cached), with the matching block of the config
    assert calls == [((120, 200, 3), (120, 200, 3), 250000, 0.55), ((120, 200, 3), (300, 300, 3), 250000, 0.55)]
    cache = (tmp_path / "ORTHOPHOTOS" / "master_frames" / "U.txt").read_text().splitlines()
    np.testing.assert_allclose(np.array(cache[0].split(","), float).reshape(3, 3), H_mo)
    assert cache[2] == "# Hash of the master frame" and cache[3] == "Hash: " + gs.compute_hash(np.full((120, 200, 3), 85, np.uint8))
    assert cache[6].startswith("Stats: Keypoints in master frame: 5000, in ortho: 6000. Inliers: 400 out of 900 matches")
    Hw = np.loadtxt(src.parent / "results" / "U_clip_geo_transf.txt", delimiter=",").reshape(3, 3)
    np.testing.assert_allclose(Hw, H_mo @ H_rm)
    df = pd.read_csv(src.parent / "results" / "U_clip.csv")
    assert list(df.columns) == ["Vehicle_ID", "Timestamp", "Frame_Number", "Ortho_X", "Ortho_Y", "Local_X", "Local_Y", "Latitude", "Longitude",
                                "Vehicle_Length", "Vehicle_Width", "Vehicle_Class", "Vehicle_Speed", "Vehicle_Acceleration", "Road_Section",
                                "Lane_Number", "Visibility"]
    assert len(df) == 6 * 30 and df["Vehicle_ID"].nunique() == 6
    r0 = df.iloc[0]
    assert r0["Ortho_X"] == pytest.approx(1.5 * (25 + 0.1 + 2) + 10, abs=0.051) and r0["Ortho_Y"] == pytest.approx(1.5 * (40 - 0.1 + 3) + 20, abs=0.051)
    assert r0["Timestamp"] == "2022-10-04 10:00:00.000" and df.iloc[29]["Timestamp"] == "0000-00-00 00:00:00.000"   # beyond the flight log
    assert set(df["Road_Section"].dropna()) == {"A"} and set(df["Lane_Number"].dropna().astype(int)) <= {1, 2}
    assert df["Vehicle_Speed"].notna().sum() > 100
    # second run: the cached master -> orthophoto homography is used (one registration only) ...
    calls.clear()
    gs.georeference(_args(src), logger)
    assert len(calls) == 1
    # ... unless the master frame changed (hash differs) or --recompute is given
    np.save(tmp_path / "ORTHOPHOTOS" / "master_frames" / "U.npy", np.full((120, 200, 3), 86, np.uint8))
    calls.clear()
    gs.georeference(_args(src), logger)
    assert len(calls) == 2
    calls.clear()
    gs.georeference(_args(src, recompute=True), logger)
    assert len(calls) == 2
    # --no-master: one direct registration against the orthophoto
    calls.clear()
    gs.georeference(_args(src, no_master=True), logger)
    assert calls == [((120, 200, 3), (300, 300, 3), 250000, 0.55)]
    # no tracks file -> the stage exits like the reference
    (src.parent / "results" / "U_clip.txt").unlink()
    with pytest.raises(SystemExit):
        gs.georeference(_args(src), logger)


def test_cli_flags_match_the_reference():
    from geotrax_amd import georef_stage as gs

    a = gs.parse_cli_args(["U.npy", "-orf", "O", "-gs", "text-file", "-rf", "3", "-nm", "-mf", "M", "-r", "-osf", "S", "-c", "default", "-of", "out"])
    assert (a.ortho_folder, a.geo_source, a.ref_frame, a.no_master, a.master_folder, a.recompute, a.segmentation_folder, a.output_folder) == \
           (Path("O"), "text-file", 3, True, Path("M"), True, Path("S"), "out")
    d = gs.parse_cli_args(["U.npy"])
    assert all(getattr(d, k) is None for k in ("ortho_folder", "geo_source", "ref_frame", "no_master", "master_folder", "recompute", "segmentation_folder"))
