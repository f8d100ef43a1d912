"""The GMC (BoT-SORT's sparse-optical-flow camera-motion estimate) alone on the GPU: under
`rocprofv3 --kernel-trace --stats -- python3 tools/gmc_profile.py` the duration of each of its kernels when nothing competes for
CU slots. Usage: python tools/gmc_profile.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
sys.path.insert(0, ROOT)
from geotrax_amd import _lib  # noqa: E402
from geotrax_amd.gmc import GMC  # noqa: E402
from geotrax_amd.synth import make_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = _lib.default_context(0)
sc = make_scene(seed=0, h=2160, w=3840)
frames = [sc.render(t, 150) for t in (0, 3, 6, 9)]
g = GMC((2160, 3840), ctx=ctx)
for k in range(4):
    g.apply(frames[k % 4])
t0 = time.perf_counter()
for k in range(n):
    g.apply(frames[k % 4])
dt = time.perf_counter() - t0
print(f"{n} blocking apply() calls on host frames: {1000 * dt / n:.2f} ms per frame wall (frame upload included)")
