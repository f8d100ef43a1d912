"""The CLI / config surface of the extract stage:
preset names for --cfg (cli_utils.py:16-32, config_utils.py:38-63), --log-path (logging_utils.py:75-110),
the run-metadata sections (extract.py:526-568) and the flag spelling (extract.py:571-603). CPU only."""
import argparse
import logging
from pathlib import Path

import pytest
import yaml

logger = logging.getLogger("cfg-test")


def test_preset_names_resolve_and_differ_as_the_reference_presets_do():
    from geotrax_amd.config_utils import CFG_DIR, PRESETS, load_config, resolve_config_path

    for name in PRESETS:
        for spelling in (name, f"{name}.yaml", f"cfg/{name}.yaml", f"geotrax/cfg/{name}.yaml"):
            assert resolve_config_path(spelling) == CFG_DIR / f"{name}.yaml", spelling
    d, c, l, s = (load_config(n, logger) for n in PRESETS)
    # the differences the reference's preset headers list (geotrax/cfg/{confident,lenient,stable}.yaml)
    assert (d["ultralytics"]["conf"], d["ultralytics"]["iou"], d["extraction"]["min_track_length"]) == (0.25, 0.7, 3)
    assert (c["ultralytics"]["conf"], c["ultralytics"]["iou"], c["extraction"]["min_track_length"]) == (0.4, 0.6, 8)
    assert (l["ultralytics"]["conf"], l["ultralytics"]["iou"], l["ultralytics"]["max_det"]) == (0.15, 0.8, 1500)
    b = l["tracker"]["botsort"]
    assert (b["track_high_thresh"], b["new_track_thresh"], b["track_buffer"], b["match_thresh"]) == (0.2, 0.1, 45, 0.9)
    assert l["tracker"]["bytetrack"]["track_buffer"] == 30 and b["gmc_method"] == "sparseOptFlow"     # untouched keys come from the base
    assert (s["stabilo"]["clahe"], s["stabilo"]["downsample_ratio"], s["stabilo"]["max_features"], s["stabilo"]["filter_ratio"]) == (True, 1.0, 4000, 0.8)
    assert c["stabilo"] == d["stabilo"] and "_base" not in c


def test_a_full_config_file_and_an_overlay_file_load(tmp_path):
    from geotrax_amd.config_utils import DEFAULT_CFG, load_config, load_config_all

    full = yaml.safe_load(DEFAULT_CFG.read_text())
    full["ultralytics"]["conf"] = 0.33
    full["visualization"] = {"mode": 1}
    p = tmp_path / "mine.yaml"
    p.write_text(yaml.safe_dump(full))
    assert load_config(p, logger)["ultralytics"]["conf"] == 0.33
    q = tmp_path / "overlay.yaml"
    q.write_text(f"_base: {p}\nultralytics:\n  iou: 0.5\n")
    cfg = load_config(q, logger)
    assert (cfg["ultralytics"]["conf"], cfg["ultralytics"]["iou"]) == (0.33, 0.5)
    args = argparse.Namespace(cfg=p, model=["synthetic:0"], class_names=None, classes=None, conf=0.5, show=None)
    allc = load_config_all(args, logger, model_names={0: "car"})
    assert allc["ultralytics"]["conf"] == 0.5 and allc["main"]["tracker_active"] == "botsort"       # CLI over config
    with pytest.raises(SystemExit):
        load_config(tmp_path / "missing.yaml", logger)


def test_cli_flags_log_path_and_metadata_sections(tmp_path):
    from geotrax_amd import extract as ex

    a = ex.parse_cli_args(["clip.npy", "-c", "confident", "-of", "out", "-lp", str(tmp_path), "-v", "-m", "w.safetensors", "-cn", "0=car",
                           "-co", "0.3", "-cls", "0", "2", "-cfl", "5", "-cfr", "50", "--interpolate"])
    assert (str(a.cfg), a.output_folder, a.log_path, a.verbose, a.model, a.class_names, a.conf, a.classes, a.cut_frame_left, a.cut_frame_right,
            a.interpolate) == ("confident", "out", tmp_path, True, ["w.safetensors"], ["0=car"], 0.3, [0, 2], 5, 50, True)
    d = ex.parse_cli_args(["clip.npy"])                        # every processing flag defaults to None and is back-filled from the YAML
    assert all(getattr(d, k) is None for k in ("model", "class_names", "conf", "classes", "cut_frame_left", "cut_frame_right", "interpolate", "log_path"))
    # --log-path: a directory gets <stage>.log, a file path is used as is
    lg = ex.setup_logger("geotrax_amd.extract", verbose=False, log_path=tmp_path)
    lg.info("to the file only")
    lg.warning("to both")
    for h in lg.handlers:
        h.flush()
    text = (tmp_path / "extract.log").read_text()
    assert "to the file only" in text and "to both" in text
    f = tmp_path / "sub" / "custom.log"
    lg = ex.setup_logger("geotrax_amd.extract", log_path=f)
    lg.info("hello")
    for h in lg.handlers:
        h.flush()
    assert "hello" in f.read_text()
    ex.setup_logger("geotrax_amd.extract", dry_run=True)       # drops the file handlers again
    # run metadata: the reference's fourteen sections, in its order (extract.py:534-567)
    from geotrax_amd.config_utils import load_config_all

    args = argparse.Namespace(source="x.npy", cfg=None, model=["synthetic:0"], class_names=None, classes=None, conf=None, show=None,
                              ortho_folder="ORTHO", master_folder=None)
    meta = ex._build_run_metadata(load_config_all(args, logger, model_names={0: "car", 1: "bus", 2: "truck", 3: "motorcycle"}), Path("results"))
    assert list(meta) == ["run", "model", "class_names", "extraction", "processing", "output", "detection", "tracker", "stabilo", "georef",
                          "paths", "visualization", "plotting", "batch"]
    assert meta["paths"] == {"ortho_folder": "ORTHO", "master_folder": None, "segmentation_folder": None}
    assert meta["class_names"]["mapping"] == {0: "car", 1: "bus", 2: "truck", 3: "motorcycle"} and meta["tracker"]["active"] == "botsort"


def test_inference_options_that_would_change_the_detections_are_refused_not_ignored():
    """ultralytics.augment (default.yaml:243) selects test-time augmentation: a different set of detections. The host class
    raises instead of silently running the plain forward pass."""
    import numpy as np
    import pytest
    from geotrax_amd.model import YOLO

    m = YOLO.__new__(YOLO)
    m._det = m._det_key = None
    with pytest.raises(NotImplementedError):
        m._detector((64, 64), {"augment": True, "imgsz": 64})


def test_a_config_without_rect_runs_the_square_letterbox_like_the_reference_default():
    """default.yaml:300 says `rect: false` (1920x1920 network input); a hand-written config that leaves the key out must not
    silently run the 1088x1920 rectangle (VERDICT r03 item 17 / SURVEY R1)."""
    from geotrax_amd.extract import _engine_kwargs

    base = {"main": {"extraction": {"stabilize": True}, "engine": {}}, "stabilo": {}, "ultralytics": {"imgsz": 1920}}
    assert _engine_kwargs(base)[0]["rect"] is False
    base["ultralytics"]["rect"] = True
    assert _engine_kwargs(base)[0]["rect"] is True


def test_engine_priority_variable_is_parsed_leniently_and_refused_clearly():
    from geotrax_amd.engine import parse_prio

    assert parse_prio(None) == (0, 0) and parse_prio("1") == (1, 1) and parse_prio("1,-1") == (1, -1) and parse_prio(" 0 , 1 ") == (0, 1)
    for bad in ("2", "a,b", "1,2,3", "1,5"):
        try:
            parse_prio(bad)
        except ValueError as e:
            assert "GTX_ENGINE_PRIO" in str(e)
        else:
            raise AssertionError(bad)


def test_stream_to_queue_rule_reproduces_the_mappings_measured_on_mi355x():
    """queue_of_streams restates what tools/stream_map_probe.py read from rocprofv3's Queue_Id column (profiles/r04_stream_map.txt):
    eight streams; the same with the null stream opened after the first; and with it opened after the fifth."""
    from geotrax_amd.engine import queue_of_streams

    assert queue_of_streams(["c"] * 8) == [0, 1, 2, 3, 3, 2, 1, 0]
    assert queue_of_streams(["c", "n"] + ["c"] * 7) == [0, 1, 2, 3, 3, 2, 1, 0, 3]
    assert queue_of_streams(["c"] * 5 + ["n"] + ["c"] * 3) == [0, 1, 2, 3, 3, 2, 1, 0, 3]
    assert queue_of_streams(["c", "n"] + ["c"] * 11)[2:] == [2, 3, 3, 2, 1, 0, 3, 2, 1, 0, 3]


def test_the_stream_plan_gives_every_detector_a_queue_without_a_busy_neighbour():
    """Streams on one hardware queue run in order: a detector must share its queue only with streams that never run anything ('x', the
    null stream), and whatever is created next (the GMC's second stream, a priming stream) must join a queue without a detector."""
    from geotrax_amd.engine import plan_stream_order, queue_of_streams

    for n_dets, n_stab in [(2, 4), (1, 4), (1, 2), (2, 0), (1, 0), (2, 6), (2, 1)]:
        order = plan_stream_order(n_dets, n_stab)
        assert order.count("d") == n_dets and order.count("s") == n_stab and order.count("f") == 1 and order.count("g") == 1 and order.count("n") == 1
        queues = queue_of_streams(order)
        det_queues = {q for t, q in zip(order, queues) if t == "d"}
        assert len(det_queues) == n_dets, order
        for t, q in zip(order, queues):
            assert t in ("d", "x", "n") or q not in det_queues, (order, queues)
        assert queue_of_streams(order + ["next"])[-1] not in det_queues, order
        assert queue_of_streams(order + ["next", "next"])[-1] not in det_queues, order
    assert plan_stream_order(2, 4) == "d,d,s,s,s,s,n,x,f,g,x,x".split(",")
    assert plan_stream_order(3, 4)[:4] == ["d", "n", "d", "d"]          # not enough queues to isolate three detectors: detectors first
