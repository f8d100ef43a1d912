"""Where and when the workgroups of one split-f16x3 convolution launch ran (diagnostic build `make -C geo-trax_amd wglog`):
per launch the workgroups' lifetimes, how full the CUs' workgroup slots are over the launch, and how long a slot stays
empty between a workgroup's last instruction and the first instruction of the next workgroup on the same CU.
Usage: python tools/wg_turnover.py [batch]      -> profiles/rNN_wg_turnover.txt"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("GTX_WGLOG_LIB") or os.path.join(ROOT, "geo-trax_amd", "build", "libgtx_wglog.so")
os.environ["GTX_LIB"] = LIB
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
import numpy as np  # noqa: E402
from geotrax_amd import _lib, ops  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = _lib.default_context(0)
dbg = ctypes.CDLL(LIB)
dbg.gtx_debug_wg_log.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
CAP = 16384
LAYERS = [("h0.s1", 128, 192, 3, 1, 240, 3), ("h0.s2c", 128, 128, 3, 1, 240, 3), ("m4.m", 64, 64, 3, 1, 240, 3), ("m6.m", 128, 128, 3, 1, 120, 3),
          ("m5", 128, 256, 3, 2, 240, 2), ("m4.cv2", 256, 128, 1, 1, 240, 4), ("m9.cv2", 1024, 512, 1, 1, 60, 4)]
print(f"# batch {NB}; times from s_memrealtime (100 MHz, common to the XCDs); slots = workgroups a CU holds of this kernel (LDS / registers)")
for nm, cin, cout, k, s, h, slots in LAYERS:
    ops.conv2d_time(2, NB, h, h, cin, cout, k, s, iters=3, ctx=ctx)
    buf = (ctypes.c_ulonglong * (3 * CAP))()
    ops.conv2d_time(2, NB, h, h, cin, cout, k, s, iters=1, ctx=ctx)
    assert dbg.gtx_debug_wg_log(buf, CAP) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(CAP, 3)
    ho = (h - 1) // s + 1
    nwg = NB * ((ho + 7) // 8) * ((ho + 15) // 16) * ((cout + 63) // 64)
    # hardware blocks of the launch: 8 x the longest XCD range; surplus blocks leave before they log (their entries are stale or zero)
    t0s, t1s = a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
    live = (t1s > 0)
    tmax = t1s[live].max()
    live &= t0s > tmax - 100000                     # 1 ms: this launch only
    hw, t0, t1 = a[live, 0], t0s[live] * 10, t1s[live] * 10          # ns
    cu = ((hw >> np.uint64(32)) << np.uint64(16)) | (hw & np.uint64(0xFF00))       # xcc | se, sh, cu bits
    span = (t1.max() - t0.min()) / 1e3
    life = (t1 - t0) / 1e3
    gaps = []
    busy = 0.0
    for c in np.unique(cu):
        m = cu == c
        st, en = np.sort(t0[m]), np.sort(t1[m])
        busy += (t1[m] - t0[m]).sum()
        j = 0
        for x in st[slots:] if len(st) > slots else []:      # the first `slots` workgroups of a CU open its slots
            gaps.append((x - en[j]) / 1e3)                   # earliest finished workgroup not yet replaced
            j += 1
    ncu = len(np.unique(cu))
    g = np.array(gaps) if gaps else np.zeros(1)
    print(f"{nm:7s} {cin:4d}->{cout:4d} k{k} s{s} {h:3d}^2: {live.sum():5d} workgroups ({nwg} expected) on {ncu} CUs, launch {span:7.1f} us, "
          f"workgroup life {np.median(life):6.1f} us median ({life.min():.1f}-{life.max():.1f}), slots filled {busy / 1e3 / (ncu * slots * span):.2f} of the launch, "
          f"slot empty between two workgroups: median {np.median(g):5.1f} us, mean {g.mean():5.1f}, p90 {np.percentile(g, 90):5.1f} ({len(gaps)} refills)")
