"""RT-DETR bring-up probe: every probed layer's error against oracle/rtdetr_ref.py (no assertion), then a per-kernel profile.
usage: python tools/rt_probe.py [imgsz] [split 0/1] [profile 0/1]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "geo-trax_amd", ROOT, ROOT / "tests"):
    sys.path.insert(0, str(p))

from geotrax_amd import _lib  # noqa: E402
from geotrax_amd.detector import Detector  # noqa: E402
from geotrax_amd.weights import synthetic_rtdetr  # noqa: E402


def main():
    imgsz = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    split = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    prof = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    from test_rtdetr_gpu import MAP_LAYERS, _frame, _rel
    from oracle.rtdetr_ref import RtDetrRef, postprocess, stretch

    hw = (432, 768) if imgsz <= 640 else (2160, 3840)
    frame = _frame(0, hw)
    w = synthetic_rtdetr(seed=3, nc=4)
    ctx = _lib.default_context(0)
    t0 = time.time()
    det = Detector(w, hw, imgsz=imgsz, conf=0.3, max_det=300, fp32_split=split, ctx=ctx, max_batch=2)
    print(f"detector built in {time.time() - t0:.1f} s", flush=True)
    got = det.detect(frame)
    print("detections", len(got), "saturated", det.saturated(), flush=True)
    if imgsz <= 960:
        ref = RtDetrRef(w)
        t0 = time.time()
        pred = ref.forward(stretch(frame, imgsz))[0].numpy()
        print(f"oracle {time.time() - t0:.1f} s")
        for name in MAP_LAYERS:
            a = det.layer_output(name)
            r = ref.acts[name][0].permute(1, 2, 0).numpy()
            print(f"{name:28s} {a.shape} rel {_rel(a, r):.2e}  max {np.abs(r).max():.3g}", flush=True)
        for name in ("model.11.src", "model.11.q", "model.11.qkv", "model.11.attn", "model.11.t1", "model.11.norm1", "model.11.ff", "model.11.t2"):
            a = det.layer_output(name)
            print(f"{name:28s} {a.shape} absmax {np.abs(a).max():.4g} mean {a.mean():.4g}")
        shapes = [(imgsz // s, imgsz // s) for s in (8, 16, 32)]
        _, valid = ref._anchors(shapes)
        feats = (ref.acts["model.28.feats"] * valid)[0].numpy()
        enc, scores = ref.acts["model.28.enc_output"][0].numpy(), ref.acts["model.28.enc_scores"][0].numpy()
        o = 0
        for l, (h, wd) in enumerate(shapes):
            for name, r in (("feats", feats), ("enc_output", enc), ("enc_scores", scores)):
                a = det.layer_output(f"model.28.{name}.{l}").reshape(h * wd, -1)[:, :r.shape[1]]
                print(f"model.28.{name}.{l:<14d} rel {_rel(a, r[o:o + h * wd]):.2e}")
            o += h * wd
        idx = det.layer_output_int("model.28.topk").ravel()
        want = ref.topk[0].numpy()
        print("topk same", (idx == want).mean(), idx[:8], want[:8])
        same = idx == want
        emb = det.layer_output("model.28.embed")[0]
        print("embed rel", _rel(emb[same], ref.acts["model.28.enc_output"][0].numpy()[want][same]))
        for i in range(ref.ndl):
            a = det.layer_output(f"model.28.decoder.layers.{i}")[0]
            r = ref.acts[f"model.28.decoder.layers.{i}"][0].numpy()
            print(f"decoder.{i} rel {_rel(a[same], r[same]):.2e}")
        raw = det.raw_output()
        print("raw box err", np.abs(raw[same, :4] - pred[same, :4]).max(), "score err", np.abs(raw[same, 4:] - pred[same, 4:]).max())
        xyxy, score, cls, _ = postprocess(pred, frame.shape[:2], 0.3, None, 300)
        print("post:", len(got), len(score), "conf err", np.abs(got.conf - score).max() if len(got) == len(score) else None)
    if prof:
        for nb in (1, 2):
            rows = det.profile(nb=nb, iters=5)
            tot = sum(r["total_ms"] for r in rows) / 5
            print(f"--- profile nb={nb}: {tot:.3f} ms per pass, {tot / nb:.3f} ms per frame")
            for r in sorted(rows, key=lambda r: -r["total_ms"]):
                ms = r["total_ms"] / 5
                print(f"{r['kernel']:48s} {r['launches'] // 5:4d} launches {ms:8.3f} ms {r['flops'] / 5 / ms / 1e9 if ms else 0:9.1f} TFLOP/s {r['bytes'] / 5 / ms / 1e9 if ms else 0:8.2f} TB/s")
    det.close()


if __name__ == "__main__":
    main()
