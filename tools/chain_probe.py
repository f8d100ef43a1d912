"""One detector, single-frame passes at 3840x2160 / 1920x1920 (BASELINE configs[1]), for a rocprofv3 kernel trace of the
launch-by-launch order (and, at commit db87c7b, of chain mode: GTX_CONV_CHAIN=0 / 1 -- profiles/r04_chain_mode.txt):
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/chain_probe.py [passes]
    python tools/chain_probe.py --read DIR        -> one pass as a table: start offset, duration, gap to the previous end, queue
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    f = sorted(glob.glob(sys.argv[2] + "/**/*_kernel_trace.csv", recursive=True))[-1]
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
    pre = [i for i, e in enumerate(ev) if "preprocess" in e[2]]
    a, b = pre[-3], pre[-2]                                   # a late pass, first kernel = its preprocess
    t0, last_end = ev[a][0], ev[a][0]
    print(f"# {f}: pass of {b - a} kernels, {(ev[b][0] - t0) / 1e3:.1f} us from its preprocess to the next pass's")
    for s, e, n, q in ev[a:b]:
        short = n.split("(")[0].replace("gtx::(anonymous namespace)::", "").replace("void ", "")[:58]
        print(f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f}  gap {(s - last_end) / 1e3:6.1f}  q{q}  {short}")
        last_end = max(last_end, e)
    sys.exit(0)

from geotrax_amd import _lib  # noqa: E402
from geotrax_amd.detector import Detector  # noqa: E402
from geotrax_amd.synth import make_scene  # noqa: E402
from geotrax_amd.weights import synthetic_yolov8  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = _lib.default_context(0)
frame = make_scene(seed=0, h=2160, w=3840).render(0, 150)
w = synthetic_yolov8(seed=0, nc=4, scale="s")
det = Detector(w, (2160, 3840), imgsz=1920, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True, half=False, rect=False, max_batch=1, ctx=ctx)
p = ctx.dev_alloc(frame.nbytes)
ctx.dev_upload(p, frame)
import time  # noqa: E402
for _ in range(5):
    det.detect_dev(p, 1)
t0 = time.perf_counter()
for _ in range(n):
    det.detect_dev(p, 1)
print(f"{1e3 * (time.perf_counter() - t0) / n:.3f} ms per blocking single-frame pass (GTX_CONV_CHAIN={os.environ.get('GTX_CONV_CHAIN', 'auto')})")
