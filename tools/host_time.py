"""How much of the extract loop is host work vs waiting on the GPU: wraps the blocking/launching calls with timers."""
import sys, os, time, collections
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'geo-trax_amd')); sys.path.insert(0,ROOT)
import numpy as np
from geotrax_amd import _lib, detector, stabilizer, tracker, engine
T=collections.defaultdict(float); N=collections.Counter()
def wrap(cls,name,key=None):
    f=getattr(cls,name); key=key or f"{cls.__name__}.{name}"
    def g(*a,**k):
        t=time.perf_counter(); r=f(*a,**k); T[key]+=time.perf_counter()-t; N[key]+=1; return r
    setattr(cls,name,g)
wrap(detector.Detector,'collect'); wrap(detector.Detector,'submit_dev')
wrap(stabilizer.Stabilizer,'collect'); wrap(stabilizer.Stabilizer,'submit_gray_dev'); wrap(stabilizer.Stabilizer,'get_cur_trans_matrix')
wrap(tracker.Tracker,'update')
from geotrax_amd import gmc as gmcmod
wrap(gmcmod.GMC,'collect'); wrap(gmcmod.GMC,'submit_gray_dev')
lib=_lib.load()
def wrapc(name):
    f=getattr(lib,name)
    def g(*a):
        t=time.perf_counter(); r=f(*a); T['C:'+name]+=time.perf_counter()-t; N['C:'+name]+=1; return r
    setattr(lib,name,g)
for nm in ('gtx_tracker_update','gtx_detector_submit_dev','gtx_detector_collect','gtx_stabilizer_submit_gray_dev','gtx_stabilizer_collect'): wrapc(nm)
sys.argv=['bench.py','--no-cpu-baseline','--no-profile','--steps','200']+sys.argv[1:]
import bench
t0=time.perf_counter(); bench.main(); tot=time.perf_counter()-t0
print("total wall incl setup %.2fs"%tot)
for k,v in sorted(T.items(), key=lambda kv:-kv[1]): print(f"{k:40s} {v*1e3:9.1f} ms  {N[k]:6d} calls  {v/N[k]*1e6:8.1f} us/call")
