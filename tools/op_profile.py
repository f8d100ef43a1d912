"""Per-kernel-family time of the detector's forward pass with each launch alone on its stream (gtx_detector_profile:
HIP events around every launch, one pass at a time). Usage: python tools/op_profile.py [batch] [f16|f32|f32s] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'geo-trax_amd')); sys.path.insert(0, ROOT)
import numpy as np
from geotrax_amd import _lib
from geotrax_amd.detector import Detector
from geotrax_amd.weights import synthetic_yolov8
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prec = sys.argv[2] if len(sys.argv) > 2 else 'f32s'
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = _lib.default_context(0)
w = synthetic_yolov8(seed=0, nc=4)
det = Detector(w, (2160, 3840), imgsz=1920, half=prec == 'f16', fp32_split=prec == 'f32s', rect=False, max_batch=nb, ctx=ctx)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (nb, 2160, 3840, 3), dtype=np.uint8)
p = ctx.dev_alloc(frames.nbytes); ctx.dev_upload(p, frames)
det.detect_dev(p, nb)
det.profile(nb, 2)
tot = 0
for f in sorted(det.profile(nb, iters), key=(lambda d: d["kernel"]) if os.environ.get("GTX_PROFILE_PER_OP") else (lambda d: -d["total_ms"])):
    us = 1000 * f['total_ms'] / f['launches']
    tot += f['total_ms'] / iters
    print(f"{f['kernel']:50s} {f['launches'] // iters:3d} launches/pass {us:8.1f} us avg {f['flops'] / max(f['total_ms'], 1e-9) / 1e9:8.1f} TF/s {f['bytes'] / max(f['total_ms'], 1e-9) / 1e6:8.0f} GB/s")
print(f"forward pass, launches back to back on one stream: {tot:.3f} ms for {nb} frame(s)")
