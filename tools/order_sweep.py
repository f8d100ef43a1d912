"""Sweep of the engine's stream-creation order on the PRODUCT path (track_with_model on a .y4m through the read-ahead feeder): HIP
deals streams to 4 hardware queues in an order this tool does not try to model, so every distinct order of the tokens
(d = detector, s = stabilizer, f = feeder copy stream, optionally x = spare) is simply measured, each in a fresh process.
    python tools/order_sweep.py [--frames 150] [--tracker bytetrack] [--spares 0] [--repeat 1] > profiles/rNN_order_sweep.txt"""
import argparse
import itertools
import json
import logging
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "geo-trax_amd"))
sys.path.insert(0, str(ROOT))


def child(d: Path) -> None:
    from geotrax_amd import extract as ex
    from geotrax_amd.config_utils import load_config_all

    logger = logging.getLogger("order_sweep")
    logger.setLevel(logging.CRITICAL)
    a = argparse.Namespace(source=str(d / "clip.y4m"), cfg=d / "cfg.yaml", output_folder=None, log_path=None, verbose=False, model=None, class_names=None,
                           conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
    model = ex.load_detector(a, logger)
    config = load_config_all(a, logger, model_names=model.names)
    best = 0.0
    for _ in range(3):
        ex.track_with_model(model, config, logger)
        best = max(best, model.last_run["loop_fps"])
    print(json.dumps({"fps": best}))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--tracker", default="bytetrack")
    ap.add_argument("--spares", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--orders", default=None, help="comma-separated orders joined by ';' instead of every permutation")
    args = ap.parse_args()
    if args.child:
        return child(Path(args.child))
    import numpy as np
    import yaml
    from geotrax_amd import _lib
    from geotrax_amd.config_utils import DEFAULT_CFG
    from geotrax_amd.detector import Detector
    from geotrax_amd.frames import bgr_to_i420, write_y4m
    from geotrax_amd.synth import make_scene
    from geotrax_amd.weights import calibrate_cls_bias, save_weights, synthetic_yolov8

    d = Path(tempfile.mkdtemp(prefix="gtx_order_sweep_"))
    sc = make_scene(seed=0, h=2160, w=3840)
    fr = [sc.render(t, 150) for t in range(6)]
    order = list(range(6)) + list(range(4, 0, -1))
    pl = [bgr_to_i420(f) for f in fr]
    write_y4m(d / "clip.y4m", [fr[0]] + [pl[order[k % len(order)]] for k in range(1, args.frames)])
    # the bench's weights: class bias calibrated to ~132 boxes per frame
    sys.path.insert(0, str(ROOT))
    import bench

    ctx = _lib.Context(0)
    base = synthetic_yolov8(**bench.SYNTH_KW)
    kw = dict(imgsz=1920, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True, half=False, rect=False, ctx=ctx)
    det = Detector(base, (2160, 3840), **kw)
    det.detect(fr[0])
    w = calibrate_cls_bias(base, det.raw_output(logits=True)[:, 4:], 0.25, 4 * 132)
    det.close()
    ctx.close()
    save_weights(w, d / "w.safetensors")
    cfg = yaml.safe_load(DEFAULT_CFG.read_text())
    cfg["ultralytics"].update(imgsz=1920, max_det=1000, conf=0.25, classes=[0, 1, 2, 3], agnostic_nms=True, rect=False)
    cfg["tracker"]["active"] = args.tracker
    cfg["extraction"]["model"] = str(d / "w.safetensors")
    (d / "cfg.yaml").write_text(yaml.safe_dump(cfg))
    if args.orders:
        orders = [o for o in args.orders.split(";") if o]
    else:
        toks = ["d", "d", "f", "s", "s", "s", "s"] + ["x"] * args.spares + (["g"] if args.tracker in ("botsort", "deepocsort") else [])
        orders = sorted({",".join(p) for p in itertools.permutations(toks) if p[0] in ("d", "f", "x")})   # a stabilizer first changes nothing new: skip
    print(f"# {len(orders)} orders, {args.frames}-frame 3840x2160 .y4m, {args.tracker}, loop-only frames/s (best of 3 passes in a fresh process per run)", flush=True)
    res = []
    for o in orders:
        vals = []
        for _ in range(args.repeat):
            p = subprocess.run([sys.executable, __file__, "--child", str(d)], capture_output=True, text=True, timeout=300, env={**os.environ, "GTX_ENGINE_ORDER": o})
            try:
                vals.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])["fps"])
            except Exception:
                vals.append(float("nan"))
        res.append((o, vals))
        print(f"{o:24s} " + " ".join(f"{v:7.1f}" for v in vals), flush=True)
    res.sort(key=lambda r: -float(np.nanmean(r[1])))
    print("# best ten")
    for o, vals in res[:10]:
        print(f"# {o:24s} " + " ".join(f"{v:7.1f}" for v in vals))


if __name__ == "__main__":
    main()
