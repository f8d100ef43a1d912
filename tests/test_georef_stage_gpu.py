"""BASELINE configs[3] as one chain on the GPU: tracks txt -> estimate_homography (RootSIFT + 2-NN + RANSAC, HIP) ->
gtx_op_georef_points -> kinematics -> CSV + _geo_transf.txt (geotrax_amd.georef_stage.georeference; reference
geotrax/georeference.py:109-202) on a synthetic scene whose frame -> orthophoto homography is known."""
import argparse
import logging
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
logger = logging.getLogger("georef-stage-gpu")
H, W = 1080, 1920
ORTHO = (126.6412, 37.3951, 2.4e-7, -1.9e-7)


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    from PIL import Image

    from geotrax_amd.synth import make_scene

    root = tmp_path_factory.mktemp("georef")
    sc = make_scene(seed=2, h=H, w=W)
    n_frames = 24
    src = root / "DATASET" / "U_clip.npy"
    src.parent.mkdir(parents=True)
    np.save(src, np.stack([sc.render(t, 150) for t in range(2)]))
    rows = []
    for t in range(n_frames):
        cam = sc.boxes(t, 150)
        for k, ((x, y, w, h), v) in enumerate(zip(sc.veh_xywh, sc.veh_vel)):
            xs, ys = x + v[0] * t, y + v[1] * t                       # frame-0 (stabilized) coordinates: the truth
            rows.append([t, k + 1, *cam[k], xs, ys, w, h, k % 4, 0.9, max(w, h), min(w, h)])
    (src.parent / "results").mkdir()
    np.savetxt(src.parent / "results" / "U_clip.txt", np.asarray(rows), fmt="%.16g", delimiter=",")
    of = root / "ORTHOPHOTOS"
    (of / "master_frames").mkdir(parents=True)
    ortho, A = sc.orthophoto(size=2600, scale=1.15, angle=0.2)
    Image.fromarray(ortho[:, :, ::-1]).save(of / "U.png")                # RGB on disk, like any PNG
    Image.fromarray(sc.render(6, 150)[:, :, ::-1]).save(of / "master_frames" / "U.png")
    (of / "U.txt").write_text(" ".join(str(v) for v in ORTHO) + "\n")
    return src, sc, A, np.asarray(rows)


def _args(source, **over):
    a = argparse.Namespace(source=Path(source), cfg=None, output_folder=None, log_path=None, verbose=False, ortho_folder=None, geo_source=None,
                           ref_frame=None, no_master=None, master_folder=None, recompute=None, segmentation_folder=None)
    for k, v in over.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("no_master", [True, None])
def test_georeference_stage_end_to_end(gtx_ctx, tree, no_master):
    import pandas as pd

    from geotrax_amd import georef_stage as gs
    from geotrax_amd import georeference as G

    src, sc, A, rows = tree
    gs.georeference(_args(src, no_master=no_master), logger, ctx=gtx_ctx)
    Hw = np.loadtxt(src.parent / "results" / "U_clip_geo_transf.txt", delimiter=",").reshape(3, 3)
    ys, xs = np.meshgrid(np.linspace(0, H - 1, 9), np.linspace(0, W - 1, 16), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    pa, pb = Hw @ P, A @ P
    err = np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()
    assert err < (0.3 if no_master else 0.6), err                    # via the master frame two registrations chain
    df = pd.read_csv(src.parent / "results" / "U_clip.csv")
    assert list(df.columns)[:3] == ["Vehicle_ID", "Frame_Number", "Ortho_X"] and len(df) == len(rows)   # no flight log -> no Timestamp column
    ox, oy = G.apply_homography(rows[:, 6], rows[:, 7], A)
    assert np.abs(df["Ortho_X"].to_numpy() - ox).max() < 1.0 and np.abs(df["Ortho_Y"].to_numpy() - oy).max() < 1.0
    lat, lon = G.ortho2geo(ox, oy, ORTHO + (0.0, 0.0))
    xl, yl = G.geo2local(lat, lon, "epsg:4326", "epsg:5186")
    assert np.abs(df["Local_X"].to_numpy() - xl).max() < 0.05 and np.abs(df["Local_Y"].to_numpy() - yl).max() < 0.05     # 1 px ~ 2 cm here
    assert df["Vehicle_Speed"].notna().sum() > 0.5 * len(df) and (df["Vehicle_Speed"].dropna() < 40).all()
    if not no_master:
        cache = src.parent.parent / "ORTHOPHOTOS" / "master_frames" / "U.txt"
        first = cache.read_text()
        gs.georeference(_args(src), logger, ctx=gtx_ctx)                 # cached master -> orthophoto homography is reused
        assert cache.read_text() == first


def test_batch_chains_extract_and_georeference(gtx_ctx, tmp_path):
    """geotrax_amd.batch on a directory (batch_process.py:288-307): extract then georeference per video on the GPU, each stage
    behind its own skip rule; a second run skips both, --geo-only --overwrite re-runs the second stage only."""
    import yaml
    from PIL import Image

    from geotrax_amd import batch
    from geotrax_amd.config_utils import DEFAULT_CFG
    from geotrax_amd.detector import Detector
    from geotrax_amd.synth import make_scene
    from geotrax_amd.weights import calibrate_cls_bias, save_weights, synthetic_yolov8

    sc = make_scene(seed=2, h=H, w=W)
    src = tmp_path / "DATASET" / "U_clip.npy"
    src.parent.mkdir(parents=True)
    frames = np.stack([sc.render(t, 150) for t in range(6)])
    np.save(src, frames)
    of = tmp_path / "ORTHOPHOTOS"
    (of / "master_frames").mkdir(parents=True)
    ortho, A = sc.orthophoto(size=2600, scale=1.15, angle=0.2)
    Image.fromarray(ortho[:, :, ::-1]).save(of / "U.png")
    Image.fromarray(sc.render(3, 150)[:, :, ::-1]).save(of / "master_frames" / "U.png")
    (of / "U.txt").write_text(" ".join(str(v) for v in ORTHO) + "\n")
    # seeded weights calibrated to a few dozen boxes per frame
    w = synthetic_yolov8(seed=1, nc=4)
    det = Detector(w, (H, W), imgsz=640, half=True, rect=True, ctx=gtx_ctx)
    det.detect(frames[0])
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 60)
    det.close()
    wpath = tmp_path / "weights.safetensors"
    save_weights(w, wpath)
    cfg = yaml.safe_load(DEFAULT_CFG.read_text())
    cfg["ultralytics"].update(imgsz=640, half=True, rect=True)
    cfg["extraction"].update(model=str(wpath), min_track_length=2)
    cfg["tracker"]["active"] = "bytetrack"
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))

    res = src.parent / "results"
    stats = {}
    counts = batch.process_input(batch.parse_cli_args([str(tmp_path / "DATASET"), "--cfg", str(cfg_path), "-y"]), logger, stats=stats)
    assert counts == dict(done=1, skipped=0, failed=0, dry=0), (counts, stats)
    assert stats["extract"]["done"] == 1 and stats["georef"]["done"] == 1
    for name in ("U_clip.txt", "U_clip_vid_transf.txt", "U_clip.csv", "U_clip_geo_transf.txt"):
        assert (res / name).exists(), name
    Hw = np.loadtxt(res / "U_clip_geo_transf.txt", delimiter=",").reshape(3, 3)
    ys, xs = np.meshgrid(np.linspace(0, H - 1, 9), np.linspace(0, W - 1, 16), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    pa, pb = Hw @ P, A @ P
    assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 1.0            # reference frame -> master -> orthophoto, two registrations
    stamp = (res / "U_clip.txt").stat().st_mtime_ns, (res / "U_clip.csv").stat().st_mtime_ns
    stats = {}
    counts = batch.process_input(batch.parse_cli_args([str(tmp_path / "DATASET"), "--cfg", str(cfg_path)]), logger, stats=stats)
    assert counts["skipped"] == 1 and stats["extract"]["skipped"] == 1 and stats["georef"]["skipped"] == 1
    stats = {}
    counts = batch.process_input(batch.parse_cli_args([str(tmp_path / "DATASET"), "--cfg", str(cfg_path), "--geo-only", "--overwrite", "-y"]), logger, stats=stats)
    assert counts["done"] == 1 and "extract" not in stats and stats["georef"]["done"] == 1
    assert (res / "U_clip.txt").stat().st_mtime_ns == stamp[0] and (res / "U_clip.csv").stat().st_mtime_ns >= stamp[1]
