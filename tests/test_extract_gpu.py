"""The whole extract path on the GPU (reader -> HIP detector -> C++ tracker -> HIP stabilizer ->
post-processing -> writers) against the same chain assembled from the oracle modules, on a short
seeded sequence, plus the output-file contract of the reference (column layout, %g / %.16g
formats, frame-0 rule, metadata file)."""
import argparse
import logging
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu
logger = logging.getLogger("test_extract")

H, W, NF, IMGSZ = 432, 768, 6, 384
STAB = dict(downsample_ratio=0.5, max_features=500, ref_multiplier=2.0, filter_ratio=0.9, ransac_epipolar_threshold=2.0,
            ransac_max_iter=5000, mask_use=True, mask_margin_ratio=0.15, detector_name="orb", matcher_name="bf",
            filter_type="ratio", transformation_type="projective", clahe=False)


def _weights_file(tmp_path, gtx_ctx, probe_frame, half=False):
    """Seeded weights whose class bias is calibrated on the probe frame so that ~60 anchors clear
    conf -- a few dozen boxes per frame, so that the foreground mask leaves the stabilizer most of
    the 384x216 working image (seeded weights know nothing about vehicles), stored as the .safetensors file + names
    side-car the model loader reads."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import calibrate_cls_bias, save_weights, synthetic_yolov8

    w = synthetic_yolov8(seed=1, nc=4)
    det = Detector(w, (H, W), imgsz=IMGSZ, half=half, rect=True, ctx=gtx_ctx)
    det.detect(probe_frame)
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 60)
    det.close()
    path = tmp_path / "weights.safetensors"
    save_weights(w, path)
    path.with_suffix(".names.yaml").write_text("{0: car, 1: bus, 2: truck, 3: motorcycle}\n")
    return path, w


def _rtdetr_weights_file(tmp_path, gtx_ctx, probe_frame):
    """Seeded RT-DETR-l weights whose final score head is shifted so that ~40 of the 300 queries clear conf on the probe frame."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import calibrate_rtdetr_scores, save_weights, synthetic_rtdetr

    w = synthetic_rtdetr(seed=3, nc=4)
    det = Detector(w, (H, W), imgsz=IMGSZ, ctx=gtx_ctx)
    det.detect(probe_frame)
    w = calibrate_rtdetr_scores(w, det.raw_output(logits=True)[:, 4:], 0.25, 40)
    det.close()
    path = tmp_path / "rtdetr-l.safetensors"
    save_weights(w, path)
    path.with_suffix(".names.yaml").write_text("{0: car, 1: bus, 2: truck, 3: motorcycle}\n")
    return path, w


def _cfg_file(tmp_path, model_path, tracker="bytetrack", half=False, gmc_method=None, with_reid=False):
    import yaml
    from geotrax_amd.config_utils import DEFAULT_CFG

    cfg = yaml.safe_load(DEFAULT_CFG.read_text())
    # rect=True: minimal letterbox padding. Seeded weights fire anywhere, also inside the grey bars of
    # a square letterbox, and such boxes clip to zero height (NaN aspect ratio in any XYAH tracker).
    cfg["ultralytics"].update(imgsz=IMGSZ, half=half, max_det=300, rect=True)
    cfg["stabilo"].update(STAB)
    cfg["tracker"]["active"] = tracker
    if tracker == "tracktrack":                                # the seeded weights' detections sit just above conf = 0.25: far below tracktrack's 0.6 / 0.7
        cfg["tracker"][tracker].update(track_high_thresh=0.25, new_track_thresh=0.25, track_low_thresh=0.1, min_track_len=2)
    if gmc_method is not None:                                 # deepocsort ships with gmc_method: none (default.yaml:421)
        cfg["tracker"][tracker]["gmc_method"] = gmc_method
        # the seeded weights are calibrated so that the detections sit just above conf = 0.25: below deepocsort's own 0.3
        cfg["tracker"][tracker].update(track_high_thresh=0.25, new_track_thresh=0.25)
    if with_reid:
        cfg["tracker"][tracker]["with_reid"] = True            # `model: auto` is the config's default (default.yaml:379)
    cfg["extraction"]["model"] = str(model_path)
    cfg["extraction"]["min_track_length"] = 2
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    return p, cfg


def _oracle_chain(frames, weights, cfg, pattern=None):
    """detect -> track -> stabilize with the oracle modules, following extract.py:145-197."""
    from oracle.bytetrack_ref import ByteTrackRef
    from oracle.stabilo_ref import StabilizerRef
    from oracle.yolov8_ref import YoloV8Ref, detect

    from oracle.gmc_ref import GmcRef
    from oracle.yolov8_ref import bgr2gray_half

    u = cfg["ultralytics"]
    from geotrax_amd.weights import is_rtdetr

    rt = is_rtdetr(weights)                 # the RT-DETR graph (extract.py:222-225): oracle/rtdetr_ref.py in the detector's place
    if rt:
        from oracle import rtdetr_ref

        model = rtdetr_ref.RtDetrRef(weights)
    else:
        model = YoloV8Ref(weights, emulate_half=False)
    active = cfg["tracker"]["active"]
    tp = cfg["tracker"][active]
    if active in ("ocsort", "deepocsort"):
        from oracle.ocsort_ref import OCSortRef

        trk = OCSortRef(cmc=(active == "deepocsort"), **{k: tp[k] for k in ("track_high_thresh", "track_low_thresh", "new_track_thresh", "track_buffer",
                                                                            "match_thresh", "delta_t", "inertia", "use_byte")},
                        **({k: tp[k] for k in ("with_reid", "proximity_thresh", "appearance_thresh", "alpha_fixed_emb")} if active == "deepocsort" else {}))
    elif active == "fasttrack":
        from oracle.fasttrack_ref import FastTrackRef

        trk = FastTrackRef(**{k: v for k, v in tp.items() if k != "tracker_type"})
    elif active == "tracktrack":
        from oracle.tracktrack_ref import TrackTrackRef

        trk = TrackTrackRef(**{k: v for k, v in tp.items() if k not in ("tracker_type", "gmc_method", "model")})
    else:
        trk = ByteTrackRef(botsort=(active == "botsort"), **{k: tp[k] for k in ("track_high_thresh", "track_low_thresh", "new_track_thresh",
                                                                                 "track_buffer", "match_thresh", "fuse_score")},
                           **({k: tp[k] for k in ("with_reid", "proximity_thresh", "appearance_thresh")} if active == "botsort" else {}))
    reid = active in ("botsort", "deepocsort", "tracktrack") and bool(tp.get("with_reid"))
    gmc = GmcRef(seed=0) if active in ("botsort", "deepocsort", "tracktrack") and tp.get("gmc_method") == "sparseOptFlow" else None
    ecc = None
    if active in ("botsort", "deepocsort", "tracktrack") and tp.get("gmc_method") == "ecc":
        from oracle.ecc_ref import EccRef

        ecc = EccRef()                                       # takes the BGR frame: the blur comes before the reduction
    scfg = dict(downsample_ratio=0.5, max_features=STAB["max_features"], ref_multiplier=2.0, filter_ratio=0.9,
                ransac_threshold=2.0, mask_use=True, mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0)
    stab = StabilizerRef(scfg, (H, W), pattern, n_hyp=2048)
    rows, transforms = [], []
    for f, frame in enumerate(frames):
        if rt:
            xyxy, conf, cls = rtdetr_ref.detect(model, frame, u["imgsz"], u["conf"], u["classes"], u["max_det"])
            feats = []
        else:
            xyxy, conf, cls, *feats = detect(model, frame, u["imgsz"], u["rect"], u["conf"], u["iou"], u["classes"], u["agnostic_nms"], u["max_det"],
                                             return_feats=reid)
        warp = gmc.apply(bgr2gray_half(frame)) if gmc is not None else (ecc.apply(frame) if ecc is not None else None)     # BOTSORT.update: camera motion first
        t = trk.update(xyxy, conf, cls, gmc=warp, **({"feats": feats[0]} if reid else {}))   # every frame, with or without detections (ultralytics track.py)
        if len(conf):
            if len(t):
                bx, ids, sc, cl = t[:, :4], t[:, 4], t[:, 5], t[:, 6]
            else:
                bx, ids, sc, cl = xyxy, np.full(len(conf), -1.0), conf, cls
            xywh = np.stack([(bx[:, 0] + bx[:, 2]) / 2, (bx[:, 1] + bx[:, 3]) / 2, bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]], 1).astype(np.float32)
        else:
            xywh = None
        if f == 0:
            stab.set_ref_frame(frame, xywh)
            sb = xywh
        else:
            Hm, _ = stab.stabilize(frame, xywh)
            if Hm is not None:
                transforms.append(np.r_[f, Hm.ravel()])
            sb = None
            if xywh is not None:
                sb = xywh.copy()
                if Hm is not None:
                    for k, (cx, cy, w, h) in enumerate(xywh.astype(np.float64)):
                        c = np.array([[cx - w / 2, cy - h / 2, 1], [cx + w / 2, cy - h / 2, 1], [cx + w / 2, cy + h / 2, 1], [cx - w / 2, cy + h / 2, 1]]).T
                        p = Hm @ c
                        p = p[:2] / p[2]
                        sb[k] = [(p[0].min() + p[0].max()) / 2, (p[1].min() + p[1].max()) / 2, np.ptp(p[0]), np.ptp(p[1])]
        if xywh is not None:
            for k in range(len(xywh)):
                rows.append([f, ids[k], *xywh[k], *sb[k], cl[k], sc[k]])
    t = np.asarray(rows, dtype=np.float32)
    return t[t[:, 1] != -1], np.asarray(transforms)


@pytest.mark.parametrize("tracker", ["bytetrack", "botsort", "botsort+reid", "botsort+ecc", "ocsort", "deepocsort", "deepocsort+reid", "fasttrack", "tracktrack", "tracktrack+reid", "rtdetr"])
def test_extract_path_matches_oracle_chain(gtx_ctx, tmp_path, tracker):
    reid = tracker.endswith("+reid")        # the appearance branch on detector-derived vectors (`with_reid: true, model: auto`): BoT-SORT, Deep OC-SORT, TrackTrack
    ecc = tracker.endswith("+ecc")          # `gmc_method: ecc` instead of BoT-SORT's default sparseOptFlow
    rtdetr = tracker == "rtdetr"            # the RT-DETR detector (a weight file with that graph) in front of ByteTrack
    tracker = "bytetrack" if rtdetr else tracker.split("+")[0]
    from geotrax_amd import extract as ex
    from geotrax_amd.config_utils import load_config_all
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene

    scene = make_scene(seed=2, h=H, w=W)
    frames = np.stack([scene.render(t, 150) for t in range(0, NF * 12, 12)])
    src = tmp_path / "clip.npy"
    np.save(src, frames)
    wpath, weights = (_rtdetr_weights_file if rtdetr else _weights_file)(tmp_path, gtx_ctx, frames[0])
    cfg_path, cfg = _cfg_file(tmp_path, wpath, tracker=tracker, gmc_method="ecc" if ecc else ("sparseOptFlow" if tracker == "deepocsort" else None), with_reid=reid)
    if tracker == "botsort" and not ecc:
        assert cfg["tracker"]["botsort"]["gmc_method"] == "sparseOptFlow"     # the reference default (default.yaml:374)
    args = argparse.Namespace(source=str(src), cfg=cfg_path, output_folder=None, log_path=None, verbose=False, model=None,
                              class_names=None, conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
    model = ex.load_detector(args, logger)
    assert ("rtdetr" in model.model.yaml_file) == rtdetr                     # what the reference reads to pick RTDETR (extract.py:223-225)
    config = load_config_all(args, logger, model_names=model.names)
    args.cut_frame_left, args.cut_frame_right = 0, None
    tracks, transforms = ex.track_with_model(model, config, logger)

    ref_tracks, ref_transforms = _oracle_chain(frames, weights, cfg)        # the oracle's own BRIEF table
    assert model.names[0] == "car"

    assert tracks.dtype == np.float32 and tracks.shape[1] == 12 and len(tracks) > 20
    assert tracks.shape == ref_tracks.shape
    np.testing.assert_array_equal(tracks[:, 0], ref_tracks[:, 0])             # frames
    # Two detections whose confidences differ by less than fp32 summation noise may swap places in
    # the NMS output and therefore swap the ids they are born with; everything else must agree.
    # Align rows per frame by box position and require the id relabelling to be a bijection.
    id_map = {}
    for f in np.unique(tracks[:, 0]):
        a, b = tracks[tracks[:, 0] == f], ref_tracks[ref_tracks[:, 0] == f]
        a, b = a[np.lexsort((a[:, 3], a[:, 2]))], b[np.lexsort((b[:, 3], b[:, 2]))]
        np.testing.assert_allclose(a[:, 2:6], b[:, 2:6], atol=2e-2)           # tracker boxes (px)
        np.testing.assert_allclose(a[:, 6:10], b[:, 6:10], atol=2e-2)         # stabilized boxes (px)
        np.testing.assert_array_equal(a[:, 10], b[:, 10])                     # class
        np.testing.assert_allclose(a[:, 11], b[:, 11], atol=1e-5)             # confidence
        for ia, ib in zip(a[:, 1], b[:, 1]):
            assert id_map.setdefault(int(ia), int(ib)) == int(ib)
    assert len(set(id_map.values())) == len(id_map)
    assert sum(k != v for k, v in id_map.items()) <= 4
    assert transforms.shape == ref_transforms.shape == (NF - 1, 10)
    np.testing.assert_array_equal(transforms[:, 0], np.arange(1, NF))
    for a, b in zip(transforms, ref_transforms):
        Ha, Hb = a[1:].reshape(3, 3), b[1:].reshape(3, 3)
        g = np.array([[0, 0, 1], [W, 0, 1], [0, H, 1], [W, H, 1.0]]).T
        pa, pb = Ha @ g, Hb @ g
        assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 1e-3
    # frame 0: stabilized boxes are the raw boxes (extract.py:178-179)
    f0 = tracks[tracks[:, 0] == 0]
    np.testing.assert_array_equal(f0[:, 2:6], f0[:, 6:10])


def test_extract_cli_writes_reference_files(gtx_ctx, tmp_path):
    import yaml
    from geotrax_amd import extract as ex

    from geotrax_amd.synth import make_scene

    wpath, _ = _weights_file(tmp_path, gtx_ctx, make_scene(seed=4, h=H, w=W).render(0), half=True)
    cfg_path, _ = _cfg_file(tmp_path, wpath, tracker="botsort", half=True)
    src = f"synthetic://?seed=4&frames=5&h={H}&w={W}"
    out = tmp_path / "out"
    import os
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        ex.main([src, "--cfg", str(cfg_path), "--output-folder", str(out), "--cut-frame-right", "3", "--interpolate"])
    finally:
        os.chdir(cwd)
    txt, tr = out / "synthetic.txt", out / "synthetic_vid_transf.txt"
    assert txt.is_file() and tr.is_file() and (tmp_path / "synthetic.yaml").is_file()
    t = np.loadtxt(txt, delimiter=",", ndmin=2)
    assert t.shape[1] == 15                                     # 14 columns + is_interpolated
    assert set(np.unique(t[:, 0])) <= {0, 1, 2, 3} and t[:, 0].max() == 3   # --cut-frame-right honoured
    first = txt.read_text().splitlines()[0].split(",")
    assert all(len(tok.replace("-", "").replace(".", "").lstrip("0")) <= 12 for tok in first)  # %g: <= 6 significant digits
    m = np.loadtxt(tr, delimiter=",", ndmin=2)
    assert m.shape == (3, 10) and list(m[:, 0]) == [1, 2, 3] and np.all(np.abs(m[:, 9] - 1) < 1e-12)
    meta = yaml.safe_load((tmp_path / "synthetic.yaml").read_text())
    assert {"run", "model", "class_names", "extraction", "processing", "output", "detection", "tracker", "stabilo", "georef"} <= set(meta)
    assert meta["tracker"]["active"] == "botsort" and meta["detection"]["imgsz"] == IMGSZ


def test_extract_cli_with_the_stable_preset(gtx_ctx, tmp_path):
    """`--cfg stable` (reference geotrax/cfg/stable.yaml: CLAHE, full-resolution stabilization, 4000 / 8000 features,
    ratio 0.8) runs end to end: the preset name resolves, the stabilizer takes the CLAHE path, transforms come out."""
    import os

    import yaml
    from geotrax_amd import extract as ex
    from geotrax_amd.synth import make_scene

    wpath, _ = _weights_file(tmp_path, gtx_ctx, make_scene(seed=4, h=H, w=W).render(0), half=False)
    src = f"synthetic://?seed=4&frames=4&h={H}&w={W}"
    out = tmp_path / "out"
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        ex.main([src, "--cfg", "stable", "--model", str(wpath), "--output-folder", str(out)])
    finally:
        os.chdir(cwd)
    tr = np.loadtxt(out / "synthetic_vid_transf.txt", delimiter=",", ndmin=2)
    assert tr.shape == (3, 10) and np.all(np.abs(tr[:, 9] - 1) < 1e-12)
    assert np.abs(tr[:, [3, 6]]).max() < 8                      # a few pixels of camera drift over 3 frames, not garbage
    meta = yaml.safe_load((tmp_path / "synthetic.yaml").read_text())
    assert meta["stabilo"]["clahe"] is True and meta["stabilo"]["downsample_ratio"] == 1.0 and meta["stabilo"]["max_features"] == 4000


def test_error_in_loop_voids_the_video(gtx_ctx, tmp_path, caplog):
    """extract.py:198-200: one exception -> logged once, empty tables, no partial output."""
    from geotrax_amd import extract as ex
    from geotrax_amd.config_utils import load_config_all

    cfg_path, _ = _cfg_file(tmp_path, "synthetic:1")
    bad = np.zeros((3, H // 2, W, 3), np.uint8)     # wrong frame height after the first frame is fine; make frame 1 differ
    frames = [np.zeros((H, W, 3), np.uint8), np.zeros((H // 2, W, 3), np.uint8)]
    d = tmp_path / "frames"
    d.mkdir()
    for i, f in enumerate(frames):
        np.save(d / f"{i:03d}.npy", f)
    args = argparse.Namespace(source=str(d), cfg=cfg_path, output_folder=None, log_path=None, verbose=False, model=None,
                              class_names=None, conf=None, classes=None, cut_frame_left=0, cut_frame_right=None, interpolate=False)
    model = ex.load_detector(args, logger)
    config = load_config_all(args, logger, model_names=model.names)
    with caplog.at_level(logging.ERROR):
        tracks, transforms = ex.track_with_model(model, config, logger)
    assert tracks.shape == (0, 12) and transforms.shape == (0, 10)
    assert any("Error processing" in r.message for r in caplog.records)
    del bad


def test_pipelined_submit_collect_equals_serial_order(gtx_ctx):
    """The asynchronous C-ABI pairs (gtx_detector_submit_dev/_collect, gtx_stabilizer_submit_gray_dev/
    _collect; two HIP streams, gray images in a 3-deep ring) must give exactly the per-frame results of
    the blocking calls, and misuse must be refused."""
    from geotrax_amd import _lib
    from geotrax_amd._lib import GtxError
    from geotrax_amd.detector import Detector
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene
    from geotrax_amd.weights import synthetic_yolov8

    hw = (720, 1280)
    sc = make_scene(seed=5, h=hw[0], w=hw[1])
    frames = [sc.render(t) for t in range(0, 60, 10)]
    from geotrax_amd.weights import calibrate_cls_bias

    w = synthetic_yolov8(seed=2, nc=4)
    det = Detector(w, hw, imgsz=640, half=True, rect=True, ctx=gtx_ctx, conf=0.25, max_det=300)
    det.detect(frames[0])
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 80)
    det.close()
    det = Detector(w, hw, imgsz=640, half=True, rect=True, ctx=gtx_ctx, conf=0.25, max_det=300)
    ctx2 = _lib.Context(0)
    kw = dict(max_features=600, seed=0)
    dptrs = []
    for f in frames:
        p = gtx_ctx.dev_alloc(f.nbytes)
        gtx_ctx.dev_upload(p, f)
        dptrs.append(p)
    try:
        # serial
        st = Stabilizer(hw, ctx=gtx_ctx, **kw)
        serial = []
        for i, p in enumerate(dptrs):
            d = det.detect_dev(p, 1)[0]
            g = det.gray_dptr(0)
            bx = d.xywh if len(d) else None
            if i == 0:
                st.set_ref_gray_dev(g[0], g[1], g[2], bx)
                serial.append((d, None))
            else:
                st.stabilize_gray_dev(g[0], g[1], g[2], bx)
                serial.append((d, st.get_cur_trans_matrix()))
        # pipelined: detect(t+1) in flight while stabilize(t) runs on the second stream, collected late
        st2 = Stabilizer(hw, ctx=ctx2, **kw)
        piped, pend = [], None
        det.submit_dev(dptrs[0], 1)
        with pytest.raises(GtxError):
            det.submit_dev(dptrs[1], 1)                     # one batch in flight per detector
        for i in range(len(dptrs)):
            d = det.collect()[0]
            g = det.gray_dptr(0)
            if i + 1 < len(dptrs):
                det.submit_dev(dptrs[i + 1], 1)
            if pend is not None:
                st2.collect()
                piped.append((pend, st2.get_cur_trans_matrix()))
                pend = None
            bx = d.xywh if len(d) else None
            if i == 0:
                st2.set_ref_gray_dev(g[0], g[1], g[2], bx)
                piped.append((d, None))
            else:
                st2.submit_gray_dev(g[0], g[1], g[2], bx)
                pend = d
        st2.collect()
        piped.append((pend, st2.get_cur_trans_matrix()))
        with pytest.raises(GtxError):
            st2.collect()                                   # nothing in flight
        with pytest.raises(GtxError):
            det.collect()
        assert len(serial) == len(piped) == len(frames)
        assert sum(len(d) for d, _ in serial) > 0
        for (ds, Hs), (dp, Hp) in zip(serial, piped):
            np.testing.assert_array_equal(ds.xyxy, dp.xyxy)
            np.testing.assert_array_equal(ds.conf, dp.conf)
            np.testing.assert_array_equal(ds.cls, dp.cls)
            assert (Hs is None) == (Hp is None)
            if Hs is not None:
                np.testing.assert_array_equal(Hs, Hp)
    finally:
        for p in dptrs:
            gtx_ctx.dev_free(p)
        det.close()


def test_batch_processes_a_directory_and_skips_existing(gtx_ctx, tmp_path):
    """geotrax_amd.batch over two tiny clips on the GPU: results written next to each clip, second run skips."""
    from geotrax_amd import batch
    from geotrax_amd.synth import make_scene

    scene = make_scene(seed=4, h=H, w=W)
    for name, t0 in (("north/clip_a.npy", 0), ("south/clip_b.npy", 30)):
        p = tmp_path / name
        p.parent.mkdir(parents=True)
        np.save(p, np.stack([scene.render(t0 + 12 * k, 150) for k in range(3)]))
    wpath, _ = _weights_file(tmp_path, gtx_ctx, scene.render(0, 150), half=True)
    cfg_path, _ = _cfg_file(tmp_path, wpath, half=True)
    args = batch.parse_cli_args([str(tmp_path), "--cfg", str(cfg_path), "--exclude-patterns", "weights", "--no-geo"])
    counts = batch.process_input(args, logger)
    assert counts["done"] == 2 and counts["failed"] == 0
    for name in ("north/results/clip_a.txt", "south/results/clip_b.txt", "north/results/clip_a_vid_transf.txt"):
        assert (tmp_path / name).exists(), name
    assert batch.process_input(batch.parse_cli_args([str(tmp_path), "--cfg", str(cfg_path), "--no-geo"]), logger)["skipped"] == 2


@pytest.mark.parametrize("ratio", [0.5, 1.0])
def test_engine_host_frames_equal_blocking_calls(gtx_ctx, ratio):
    """ExtractEngine fed with host frames (odd batch tail, both stabilizer input paths: the detector's
    half-resolution gray in HBM for downsample_ratio 0.5, the stabilizer's own gray otherwise) against the
    blocking per-frame calls of the same objects."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import calibrate_cls_bias, synthetic_yolov8

    scene = make_scene(seed=6, h=H, w=W)
    frames = [scene.render(12 * k, 150) for k in range(5)]
    kw = dict(imgsz=IMGSZ, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=1, nc=4)
    det = Detector(w, (H, W), ctx=gtx_ctx, **kw)
    det.detect(frames[0])
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 60)
    det.close()
    stab_kw = dict(max_features=500, downsample_ratio=ratio)
    # blocking reference
    det = Detector(w, (H, W), ctx=gtx_ctx, **kw)
    trk, st = Tracker("bytetrack"), Stabilizer((H, W), ctx=gtx_ctx, **stab_kw)
    want = []
    for i, f in enumerate(frames):
        d = det.detect(f)
        bx, ids = (trk.update(d.xyxy, d.conf, d.cls)[:2]) if len(d) else (d.xyxy, np.zeros(0, np.int32))
        xywh = None if len(bx) == 0 else np.stack([(bx[:, 0] + bx[:, 2]) / 2, (bx[:, 1] + bx[:, 3]) / 2, bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]], 1).astype(np.float32)
        if i == 0:
            st.set_ref_frame(f, xywh)
            want.append((ids, xywh, None))
        else:
            st.stabilize(f, xywh)
            want.append((ids, xywh, st.get_cur_trans_matrix()))
    det.close()
    eng = ExtractEngine(w, (H, W), kw, Tracker("bytetrack"), stab_kw, batch=2, det_streams=2, stab_streams=3)
    try:
        got = list(eng.run([frames[0:2], frames[2:4], frames[4:5]]))             # last batch is short
        assert [r.index for r in got] == list(range(5))
        for r, (ids, xywh, Hm) in zip(got, want):
            np.testing.assert_array_equal(r.ids if r.ids is not None else np.zeros(0, np.int32), ids)
            if xywh is None:
                assert r.xywh is None
            else:
                np.testing.assert_array_equal(r.xywh, xywh)
            assert (r.H is None) == (Hm is None)
            if Hm is not None:
                np.testing.assert_array_equal(r.H, Hm)
        assert got[0].H is None and got[0].xywh_stab is not None and sum(r.H is not None for r in got) == 4
        with pytest.raises(ValueError):
            list(eng.run([[frames[0][:100]]]))                                     # wrong frame size
        with pytest.raises(ValueError):
            list(eng.run([frames[:3]]))                                            # more frames than the batch size
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tracker", ["bytetrack", "botsort"])
def test_engine_threaded_equals_single_thread_and_survives_an_abandoned_run(gtx_ctx, monkeypatch, tracker):
    """The three-thread host pipeline (detector / tracker / stabilizer stages) returns exactly what the
    single-thread loop returns; a run abandoned half way leaves no pass in flight, so the engine can be
    reset and used again; an error raised in a worker stage reaches the caller."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import calibrate_cls_bias, synthetic_yolov8

    scene = make_scene(seed=8, h=H, w=W)
    frames = [scene.render(6 * k, 150) for k in range(9)]
    kw = dict(imgsz=IMGSZ, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=2, nc=4)
    det = Detector(w, (H, W), ctx=gtx_ctx, **kw)
    det.detect(frames[0])
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 50)
    det.close()
    batches = [frames[i:i + 2] for i in range(0, len(frames), 2)]

    def run(threads):
        monkeypatch.setenv("GTX_ENGINE_THREADS", threads)
        eng = ExtractEngine(w, (H, W), kw, Tracker(tracker), dict(max_features=500), batch=2, det_streams=2, stab_streams=3,
                            gmc=tracker == "botsort")
        try:
            first = list(eng.run(batches))
            # abandon a run after three frames, then go again from a clean state
            eng.reset()
            it = eng.run(batches)
            for _ in range(3):
                next(it)
            it.close()
            eng.reset()
            again = list(eng.run(batches))
            return first, again
        finally:
            eng.close()

    (a1, a2), (b1, _) = run("1"), run("0")
    for got, want in ((a1, b1), (a2, a1)):
        assert [r.index for r in got] == list(range(len(frames)))
        for r, q in zip(got, want):
            for x, y in ((r.ids, q.ids), (r.xywh, q.xywh), (r.H, q.H), (r.xywh_stab, q.xywh_stab)):
                assert (x is None) == (y is None)
                if x is not None:
                    np.testing.assert_array_equal(x, y)
    assert any(r.ids is not None and len(r.ids) for r in a1) and sum(r.H is not None for r in a1) == len(frames) - 1

    monkeypatch.setenv("GTX_ENGINE_THREADS", "1")
    trk = Tracker(tracker)
    eng = ExtractEngine(w, (H, W), kw, trk, dict(max_features=500), batch=2, gmc=tracker == "botsort")
    try:
        def boom(*a, **k):
            raise RuntimeError("tracker stage failed")
        monkeypatch.setattr(trk, "update", boom)
        with pytest.raises(RuntimeError, match="tracker stage failed"):
            list(eng.run(batches))
    finally:
        eng.close()


@pytest.mark.gpu
def test_frame_sharded_botsort_gmc_equals_the_unsharded_run(gtx_ctx):
    """SURVEY.md 8e with BoT-SORT: two 'ranks' take the clip's batches alternately; each primes its GMC with the frame
    before the batch and ships the warps in its records; the tracker replayed over the records in clip order gives the
    same tracks as the single engine that sees every frame in order."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.distributed import pack_frame_record, unpack_frame_gmc, unpack_frame_record
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import calibrate_cls_bias, synthetic_yolov8

    B, n_frames, max_det = 2, 12, 300
    scene = make_scene(seed=9, h=H, w=W)
    frames = [scene.render(5 * k, 150) for k in range(n_frames)]
    kw = dict(imgsz=IMGSZ, conf=0.25, iou=0.7, max_det=max_det, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=3, nc=4)
    det = Detector(w, (H, W), ctx=gtx_ctx, **kw)
    det.detect(frames[0])
    w = calibrate_cls_bias(w, det.raw_output(logits=True)[:, 4:], 0.25, 50)
    det.close()
    fbytes = frames[0].nbytes
    pool = gtx_ctx.dev_alloc(fbytes * n_frames)
    for i, f in enumerate(frames):
        gtx_ctx.dev_upload(pool + i * fbytes, f)
    try:
        whole = ExtractEngine(w, (H, W), kw, Tracker("botsort"), None, batch=B, det_streams=2, gmc=True)
        try:
            want = list(whole.run(pool + g * B * fbytes for g in range(n_frames // B)))
        finally:
            whole.close()
        assert any(r.gmc is not None and not np.array_equal(r.gmc, np.eye(2, 3)) for r in want)        # the camera does move
        records = {}
        for rank in range(2):
            eng = ExtractEngine(w, (H, W), kw, None, None, batch=B, det_streams=2, gmc=True)
            try:
                mine = [g for g in range(n_frames // B) if g % 2 == rank]
                items = [(pool + g * B * fbytes, None if g == 0 else pool + (g * B - 1) * fbytes) for g in mine]
                got = list(eng.run(items))
                for j, r in enumerate(got):
                    frame = mine[j // B] * B + j % B
                    records[frame] = pack_frame_record(max_det, r.xyxy, r.conf, r.cls, None, r.gmc, with_gmc=True)
            finally:
                eng.close()
        trk = Tracker("botsort")
        for t in range(n_frames):
            xyxy, conf, cls, _ = unpack_frame_record(records[t], max_det)
            warp = unpack_frame_gmc(records[t])
            np.testing.assert_array_equal(warp, want[t].gmc)                       # same frame pair -> same warp, bit for bit
            bx, ids = trk.update(xyxy, conf, cls, gmc=warp)[:2]
            if want[t].ids is None:
                assert len(ids) == 0
            else:
                np.testing.assert_array_equal(ids, want[t].ids)
                np.testing.assert_array_equal(bx, want[t].xyxy)
    finally:
        gtx_ctx.dev_free(pool)


@pytest.mark.gpu
@pytest.mark.parametrize("tracker", ["bytetrack", "botsort"])
def test_engine_frames_without_detections(gtx_ctx, tracker):
    """extract.py:156-187: a frame without detections writes no rows and has no boxes to warp, but is still
    registered against the reference frame (no foreground mask). The pinned ultralytics (>=8.4.80, trackers/track.py)
    calls tracker.update on EVERY frame, so the tracker (frame counter, lost/removed ageing) and BoT-SORT's GMC see
    the empty frames too. Here every frame is empty (confidence threshold above every score)."""
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import synthetic_yolov8

    scene = make_scene(seed=4, h=H, w=W)
    frames = [scene.render(8 * k, 150) for k in range(5)]
    kw = dict(imgsz=IMGSZ, conf=0.999999, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=1, nc=4, cls_bias=-30.0)
    trk = Tracker(tracker)
    calls = []
    orig = trk.update
    trk.update = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    eng = ExtractEngine(w, (H, W), kw, trk, dict(max_features=500), batch=2, det_streams=2, stab_streams=2, gmc=tracker == "botsort")
    try:
        got = list(eng.run([frames[0:2], frames[2:4], frames[4:5]]))
    finally:
        eng.close()
    assert [r.index for r in got] == list(range(5)) and len(calls) == 5     # one tracker.update per frame, empty or not
    for i, r in enumerate(got):
        assert r.n_det == 0 and r.ids is None and r.xywh is None and r.xywh_stab is None and len(r.xyxy) == 0
        assert (r.gmc is not None) == (tracker == "botsort")            # the GMC saw the frame (BOTSORT.update -> gmc.apply)
        assert (r.H is None) == (i == 0)                     # the reference frame has no transform row; the others register
    assert all(np.isfinite(r.H).all() and abs(np.linalg.det(r.H) - 1.0) < 0.05 for r in got[1:])


@pytest.mark.gpu
def test_engines_of_one_process_run_on_the_same_planned_streams(gtx_ctx):
    """engine.StreamPlan: which streams share a hardware queue depends on every stream the process has created and destroyed
    (profiles/r04_stream_map.txt), so the engine's streams are created once, in the planned order, and every engine built afterwards
    -- the next video of a batch -- runs on the same ones: same contexts by role, none destroyed by close(), a third engine that is
    built while another is still open gets streams of its own."""
    from geotrax_amd.engine import ExtractEngine, StreamPlan
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import synthetic_yolov8

    scene = make_scene(seed=4, h=H, w=W)
    frames = [scene.render(8 * k, 150) for k in range(4)]
    kw = dict(imgsz=IMGSZ, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=1, nc=4)

    def build():
        return ExtractEngine(w, (H, W), kw, Tracker("botsort"), dict(max_features=500), batch=2, det_streams=2, stab_streams=2, gmc=True, feeder_stream=True)

    def streams(e):
        return [d.ctx for d in e.dets] + [s.ctx for s in e.stabs] + [e.gmc.ctx, e.feeder_ctx]

    a = build()
    plan = StreamPlan.get(a.device, 2, 2)
    assert a.plan is plan and all(any(c is ent[1] for ent in plan.ctxs) for c in streams(a))
    first, handles = streams(a), [c.handle.value for c in streams(a)]
    assert len({id(c) for c in first}) == len(first)
    out_a = [(r.xyxy.copy(), None if r.H is None else r.H.copy()) for r in a.run([frames[0:2], frames[2:4]])]
    a.close()
    assert [c.handle.value for c in first] == handles          # close() hands the streams back, it does not destroy them
    b = build()
    try:
        assert all(x is y for x, y in zip(streams(b), first))    # the second engine: the very same streams, role by role
        c = build()                                            # a second engine alive at the same time cannot share them
        try:
            assert not any(x is y for x in streams(c) for y in first)
        finally:
            c.close()
        out_b = [(r.xyxy.copy(), None if r.H is None else r.H.copy()) for r in b.run([frames[0:2], frames[2:4]])]
    finally:
        b.close()
    for (xa, ha), (xb, hb) in zip(out_a, out_b):
        assert np.array_equal(xa, xb) and ((ha is None and hb is None) or np.array_equal(ha, hb))


@pytest.mark.gpu
@pytest.mark.parametrize("tracker", ["bytetrack", "botsort"])
def test_a_paced_source_gets_the_same_results_sooner(gtx_ctx, tracker):
    """A live stream (batches arriving slower than the pipeline works) takes the engine's other paths: stage 1 hands a batch's results on
    before it blocks for the next batch, the stabilizer stage takes its pending frames while nothing arrives. Same frames, same order,
    same numbers as the throughput run; and a frame's result is out before the next batch exists."""
    import time

    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import synthetic_yolov8

    scene = make_scene(seed=6, h=H, w=W)
    frames = [scene.render(6 * k, 150) for k in range(12)]
    kw = dict(imgsz=IMGSZ, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=True, rect=True)
    w = synthetic_yolov8(seed=1, nc=4)
    batches = [frames[i:i + 2] for i in range(0, len(frames), 2)]

    def run(source):
        eng = ExtractEngine(w, (H, W), kw, Tracker(tracker), dict(max_features=500), batch=2, det_streams=2, stab_streams=3, gmc=tracker == "botsort")
        try:
            return [(r.index, r.xyxy.copy(), None if r.ids is None else r.ids.copy(), None if r.H is None else r.H.copy(), time.perf_counter()) for r in eng.run(source)]
        finally:
            eng.close()

    made = []

    def paced():
        for b in batches:
            time.sleep(0.06)                                   # a stream: batches exist when they arrive
            made.append(time.perf_counter())
            yield b

    from geotrax_amd.engine import PacedSource

    fast, slow = run(batches), run(PacedSource(paced()))            # the source declares itself a stream
    assert [r[0] for r in slow] == [r[0] for r in fast] == list(range(len(frames)))
    for a, b in zip(fast, slow):
        assert np.array_equal(a[1], b[1]) and ((a[2] is None and b[2] is None) or np.array_equal(a[2], b[2]))
        assert (a[3] is None and b[3] is None) or np.array_equal(a[3], b[3])
    # frames of batch k (k >= 2: the pipeline has been primed) left the engine before batch k + 1 was made
    for k in range(2, len(batches) - 1):
        assert slow[2 * k + 1][4] < made[k + 1], k



def _run_sharded_cli(tmp_path, cfg_path, clip, out, n_ranks, backend, port, extra_env=None):
    import os
    import subprocess
    import sys

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GTX_DIST_BACKEND=backend, OMP_NUM_THREADS="1",
               PYTHONPATH=str(ROOT / "geo-trax_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""), **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "geotrax_amd.extract", str(clip), "--cfg", str(cfg_path), "--output-folder", str(out)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)


@pytest.mark.parametrize("tracker,run_frames", [("botsort", None), ("bytetrack", None), ("botsort", 2)])
def test_cli_under_a_launcher_shards_the_frames_of_one_video(gtx_ctx, tmp_path, tracker, run_frames):
    """`torchrun ... -m geotrax_amd.extract <video>`: contiguous frame ranges per rank, one gather, tracker replay on rank 0
    (the product path of geotrax_amd.distributed). Two ranks share the one GPU here over gloo. Against the
    single-process run: same rows, ids, raw boxes, classes, scores (the tracker sees the same detections and, with
    BoT-SORT, the same camera-motion warps: each rank primes its GMC with the frame before its range); the stabilized
    boxes agree within the 1 px bar (shard ranks mask with the raw detections, extract.py:181 with the tracker's boxes)."""
    from geotrax_amd import extract as ex
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=4, h=H, w=W)
    frames = np.stack([sc.render(3 * t, 150) for t in range(9)])
    clip = tmp_path / "U_clip.npy"
    np.save(clip, frames)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, cfg = _cfg_file(tmp_path, wpath, tracker=tracker)
    if run_frames is not None:                           # runs of 2 frames dealt round-robin: three rounds, a GMC priming frame per run
        import yaml

        cfg["engine"] = {"shard_run_frames": run_frames}
        cfg_path.write_text(yaml.safe_dump(cfg))
    ex.main([str(clip), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / "single")])
    p = _run_sharded_cli(tmp_path, cfg_path, clip, tmp_path / "sharded", 2, "gloo", 29547)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    a = np.loadtxt(tmp_path / "single" / "U_clip.txt", delimiter=",", ndmin=2)
    b = np.loadtxt(tmp_path / "sharded" / "U_clip.txt", delimiter=",", ndmin=2)
    assert a.shape == b.shape and len(a) > 50 and set(np.unique(a[:, 0])) == set(range(9))
    np.testing.assert_array_equal(a[:, [0, 1, 10, 11]], b[:, [0, 1, 10, 11]])         # frames, ids, classes, scores
    np.testing.assert_allclose(a[:, 2:6], b[:, 2:6], rtol=0, atol=1e-3)               # tracker boxes (%g text)
    # stabilized boxes differ by the mask choice only. On this 768x432 clip (500 keypoints, seeded weights whose boxes cover a
    # large part of the frame) that moves them by up to ~1.5 px; at 4K with vehicle-sized boxes the two masks agree to 0.4 px
    # (tests/test_fullsize_gpu.py::test_shard_mode_mask_moves_the_homography_by_less_than_a_pixel)
    assert np.abs(a[:, 6:10] - b[:, 6:10]).max() < 3.0
    ta = np.loadtxt(tmp_path / "single" / "U_clip_vid_transf.txt", delimiter=",", ndmin=2)
    tb = np.loadtxt(tmp_path / "sharded" / "U_clip_vid_transf.txt", delimiter=",", ndmin=2)
    assert list(ta[:, 0]) == list(tb[:, 0]) == list(range(1, 9))
    # a rank that fails voids the video for everybody: no output, no hang (rank 1's range holds frame 7)
    p = _run_sharded_cli(tmp_path, cfg_path, clip, tmp_path / "void", 2, "gloo", 29548, {"GTX_TEST_FAIL_AT_FRAME": "7"})
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert not (tmp_path / "void" / "U_clip.txt").exists()
    assert "Error processing" in p.stderr + p.stdout


def test_cli_sharded_path_over_rccl_with_one_rank(gtx_ctx, tmp_path):
    """RCCL cannot place two ranks on one GPU, so on a one-GPU box the N > 1 collectives cannot run; what can run is the same
    code path with ONE rank on the nccl backend (GTX_FRAME_SHARDING=force): process-group init on the device, the failure
    flag's all-reduce and the record gather on device tensors, the replay thread. Output == the single-process run."""
    from geotrax_amd import extract as ex
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=4, h=H, w=W)
    frames = np.stack([sc.render(3 * t, 150) for t in range(7)])
    clip = tmp_path / "U_clip.npy"
    np.save(clip, frames)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, _ = _cfg_file(tmp_path, wpath, tracker="botsort")
    ex.main([str(clip), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / "single")])
    p = _run_sharded_cli(tmp_path, cfg_path, clip, tmp_path / "sharded", 1, "nccl", 29551, {"GTX_FRAME_SHARDING": "force"})
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    a = np.loadtxt(tmp_path / "single" / "U_clip.txt", delimiter=",", ndmin=2)
    b = np.loadtxt(tmp_path / "sharded" / "U_clip.txt", delimiter=",", ndmin=2)
    assert a.shape == b.shape and len(a) > 50
    np.testing.assert_array_equal(a[:, [0, 1, 10, 11]], b[:, [0, 1, 10, 11]])
    np.testing.assert_allclose(a[:, 2:6], b[:, 2:6], rtol=0, atol=1e-3)


def test_cli_frame_sharding_over_rccl_with_two_gpus(gtx_ctx, tmp_path):
    """The same run with one GPU per rank and the records gathered over RCCL (backend nccl). Needs two visible GPUs: the
    builder's and the driver's single-GPU boxes skip it, an 8-GPU node runs it."""
    import torch

    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        pytest.skip(f"torch.cuda.device_count() = {n_gpu}: RCCL refuses two ranks on one device, so the 2-rank nccl run needs a second GPU "
                    "(the 1-rank nccl run above and the 2- and 8-rank gloo runs cover the code path on this box)")
    from geotrax_amd import extract as ex
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=4, h=H, w=W)
    frames = np.stack([sc.render(3 * t, 150) for t in range(9)])
    clip = tmp_path / "U_clip.npy"
    np.save(clip, frames)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, _ = _cfg_file(tmp_path, wpath, tracker="botsort")
    ex.main([str(clip), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / "single")])
    p = _run_sharded_cli(tmp_path, cfg_path, clip, tmp_path / "sharded", 2, "nccl", 29549)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    a = np.loadtxt(tmp_path / "single" / "U_clip.txt", delimiter=",", ndmin=2)
    b = np.loadtxt(tmp_path / "sharded" / "U_clip.txt", delimiter=",", ndmin=2)
    np.testing.assert_array_equal(a[:, [0, 1, 10, 11]], b[:, [0, 1, 10, 11]])
