import sys,time
sys.path.insert(0,'geo-trax_amd'); sys.path.insert(0,'.')
import bench
from geotrax_amd import _lib
from geotrax_amd.synth import make_scene
class A: pass
a=A(); a.imgsz=1920; a.half=1; a.rect=0; a.batch=1
ctx=_lib.Context(0); sc=make_scene(seed=0,h=bench.H,w=bench.W); f=[sc.render(t,150) for t in range(3)]
det,_,_,_=bench.calibrated_detector(ctx,f[0],a,132)
for i in range(5): det.detect(f[i%3])
t0=time.perf_counter(); n=60
for i in range(n): det.detect(f[i%3])
el=time.perf_counter()-t0
print(f"host-buffer blocking detect: {1e3*el/n:.2f} ms/frame = {n/el:.0f} frames/s")
