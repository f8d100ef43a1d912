"""Time per iteration of the ECC motion compensation (gmc.EccGMC) on a frame pair of the synthetic clip, at two frame sizes and two
iteration caps, for both forms of the warp (`warp="exact"`: OpenCV >= 4.11, the default; "fixed": through 4.10): python tools/ecc_time.py (one GPU). DESIGN.md section 3, round 6, quotes it."""
import sys, time, numpy as np
sys.path.insert(0, 'geo-trax_amd')
from geotrax_amd import _lib
from geotrax_amd.gmc import EccGMC
from geotrax_amd.synth import make_scene
for hw in ((2160, 3840), (1080, 1920)):
    sc = make_scene(seed=3, h=hw[0], w=hw[1])
    f0, f1 = sc.render(0), sc.render(30)
    for warp, cap in (("exact", 5000), ("fixed", 100), ("fixed", 1000)):
        g = EccGMC(hw, ctx=_lib.default_context(0), max_iters=cap, warp=warp)
        g.apply(f0)
        t0 = time.perf_counter(); g.apply(f1); dt = time.perf_counter() - t0
        print(f"{hw[1]}x{hw[0]} warp {warp} cap {cap}: {g.last['iters']} iterations in {1e3 * dt:.2f} ms (host frame in) = {1e6 * dt / max(g.last['iters'], 1):.1f} us per iteration, rho {g.last['rho']:.6f}")
        g.close()
