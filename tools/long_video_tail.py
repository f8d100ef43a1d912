"""Host time of what follows the GPU on a long video: post-processing (extract.py:296-484) and the kinematics of the georeference
stage (georeference.py:705-766) on a synthetic 7 000-frame table (~950 k rows, ~3 000 tracks, some with gaps).
    python tools/long_video_tail.py [--interpolate]      (CPU only)"""
import argparse
import logging
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
from geotrax_amd import georeference as gr  # noqa: E402
from geotrax_amd import postprocess as pp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--interpolate", action="store_true")
ap.add_argument("--frames", type=int, default=7000)
a = ap.parse_args()
rng = np.random.default_rng(0)
F, life = a.frames, 300
n_tracks = int(F * 130 / life)
rows = []
for tid in range(1, n_tracks + 1):
    f0 = rng.integers(0, F - 50)
    fr = np.arange(f0, min(F, f0 + int(rng.integers(50, 2 * life))))
    if rng.random() < 0.3:
        fr = np.delete(fr, rng.integers(1, len(fr) - 1, 3))
    x = rng.uniform(200, 3600) + np.cumsum(rng.normal(2, 0.5, len(fr)))
    y = rng.uniform(200, 1900) + np.cumsum(rng.normal(0.5, 0.3, len(fr)))
    w, h = rng.uniform(40, 90) + rng.normal(0, 1, len(fr)), rng.uniform(20, 45) + rng.normal(0, 1, len(fr))
    rows.append(np.stack([fr, np.full(len(fr), tid), x, y, w, h, x + 1, y + 1, w, h, np.full(len(fr), rng.integers(0, 4)), rng.uniform(0.3, 0.95, len(fr))], 1))
tracks = np.concatenate(rows).astype(np.float32)
tracks = tracks[np.argsort(tracks[:, 0], kind="stable")]
cfg = yaml.safe_load(open(os.path.join(ROOT, "geo-trax_amd", "geotrax_amd", "cfg", "default.yaml")))
cfg["args"] = argparse.Namespace(interpolate=a.interpolate)
cfg.setdefault("tracker", {"active": "bytetrack", "bytetrack": {"track_buffer": 30}})
log = logging.getLogger("tail")
print(f"# {len(tracks)} rows, {n_tracks} tracks, interpolate={a.interpolate}")
t = time.perf_counter()
out = pp.postprocess_tracks(tracks, {"main": cfg}, log, (3840, 2160))
print(f"postprocess_tracks          {time.perf_counter() - t:6.2f} s  -> {out.shape}")
o = out[np.lexsort((out[:, 0], out[:, 1]))]
ids, fr = o[:, 1].astype(np.int64), o[:, 0].astype(np.int64)
x, y = o[:, 2].astype(np.float64) * 0.03, o[:, 3].astype(np.float64) * 0.03
interp = o[:, 14].astype(int) if o.shape[1] > 14 else None
t = time.perf_counter()
sp, ac = gr.compute_kinematics(ids, fr, x, y, rng.random(len(o)) < 0.95, 29.97, "gaussian", 14, interp)
print(f"compute_kinematics          {time.perf_counter() - t:6.2f} s  ({np.isfinite(sp).mean():.2f} of the rows get a speed)")
