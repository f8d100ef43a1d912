import sys,time
sys.path.insert(0,'geo-trax_amd'); sys.path.insert(0,'.')
import numpy as np
import bench
from geotrax_amd import _lib
from geotrax_amd.engine import ExtractEngine
from geotrax_amd.synth import make_scene
from geotrax_amd.tracker import Tracker
class A: pass
a=A(); a.imgsz=1920; a.half=1; a.rect=0; a.batch=2
H,W=bench.H,bench.W
ctx=_lib.Context(0); sc=make_scene(seed=0,h=H,w=W); fr=[sc.render(t,150) for t in range(6)]
det,weights,_,_=bench.calibrated_detector(ctx,fr[0],a,132)
kw=dict(imgsz=1920,conf=0.25,iou=0.7,max_det=1000,classes=[0,1,2,3],agnostic_nms=True,half=True,rect=False)
eng=ExtractEngine(weights,(H,W),kw,Tracker("bytetrack"),{},batch=2,detectors=[det])
def batches(n):
    for k in range(n): yield [fr[(2*k)%6], fr[(2*k+1)%6]]
list(eng.run(batches(8)))
eng.reset()
t0=time.perf_counter(); n=60; res=list(eng.run(batches(n))); el=time.perf_counter()-t0
print(f"engine from host frames: {2*n/el:.1f} fps ({1e3*el/(2*n):.2f} ms/frame), last frame tracks {len(res[-1].xyxy)}")
