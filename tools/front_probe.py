"""Phase cycles of the fused front launch (model.0 + model.1 + model.2.cv1, conv_front_split_kernel) from the diagnostic
build (`make -C geo-trax_amd stamp_front`: stamps in the front launch only): per workgroup, entry -> K loop (image loads, the
image patch -> LDS and the stem stage), the K loop, and K loop end -> stores retired. The stamps' own atomics (ten per
workgroup on eight addresses) double the launch's duration: read the proportions, not the totals.
Usage: python tools/front_probe.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.environ.get("GTX_STAMP_LIB") or os.path.join(ROOT, "geo-trax_amd", "build", "libgtx_stamp_front.so")
os.environ["GTX_LIB"] = STAMP
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from geotrax_amd import _lib  # noqa: E402
from geotrax_amd.detector import Detector  # noqa: E402
from geotrax_amd.weights import synthetic_yolov8  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = _lib.default_context(0)
dbg = ctypes.CDLL(STAMP)
dbg.gtx_debug_conv_clock.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
det = Detector(synthetic_yolov8(seed=0, nc=4), (2160, 3840), imgsz=1920, half=False, rect=False, fp32_split=True, max_batch=NB, ctx=ctx)
out = (ctypes.c_ulonglong * 8)()
det.profile(NB, 3)
dbg.gtx_debug_conv_clock(out)                        # clear
ITERS = 20
prof = det.profile(NB, ITERS)
dbg.gtx_debug_conv_clock(out)
front = [f for f in prof if f["kernel"] == "conv_front_split_kernel"][0]
c, r, n, e2l, l2x, p5, p6, p7 = list(out)
n = max(n, 1)
print(f"# batch {NB}: conv_front_split_kernel {front['total_ms'] / front['launches'] * 1000:.1f} us per launch, {n // ITERS} workgroups, in-kernel clock {c / max(r, 1) * 0.1:.2f} GHz")
print(f"# cycles per workgroup: entry -> K loop (front stage) {e2l / n:.0f} | K loop {c / n:.0f} | loop end -> stores retired {l2x / n:.0f}")
