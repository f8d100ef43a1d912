import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "geo-trax_amd"))
t0 = time.perf_counter()
import numpy as np
from geotrax_amd import _lib
from geotrax_amd.detector import Detector
from geotrax_amd.stabilizer import Stabilizer
from geotrax_amd.engine import StreamPlan
from geotrax_amd.weights import synthetic_yolov8
from geotrax_amd.feeder import FrameFeeder
t = [("imports", time.perf_counter() - t0)]
def lap(name, t1):
    t.append((name, time.perf_counter() - t1)); return time.perf_counter()
t1 = time.perf_counter()
w = synthetic_yolov8(seed=0, nc=4); t1 = lap("synthetic weights (host)", t1)
_lib.load(); t1 = lap("dlopen libgtx", t1)
plan = StreamPlan.get(0, 2, 4); t1 = lap("stream plan (HIP init + 11 streams + null)", t1)
kw = dict(imgsz=1920, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True, half=False, rect=False, max_batch=2)
d0 = Detector(w, (2160, 3840), ctx=plan.take("d"), **kw); t1 = lap("detector 0", t1)
d1 = Detector(w, (2160, 3840), ctx=plan.take("d"), **kw); t1 = lap("detector 1", t1)
ss = []
for i in range(4):
    ss.append(Stabilizer((2160, 3840), ctx=plan.take("s"))); t1 = lap(f"stabilizer {i}", t1)
fd = FrameFeeder((2160, 3840), kind="i420", batch=2, ring=6, device=0, ctx=plan.take("f")); t1 = lap("feeder (pinned + device rings)", t1)
f = np.zeros((2160, 3840, 3), np.uint8)
d0.detect(f); t1 = lap("first detect (kernel load)", t1)
d0.detect(f); t1 = lap("second detect", t1)
for n, v in t:
    print(f"{v*1e3:8.1f} ms  {n}")
