"""Clock the chip holds inside the split-f16x3 convolution's K loop (MI355X_MICROARCH.md, 'DVFS give-back' item 6).
Needs the diagnostic build (`make -C geo-trax_amd stamp` -> build/libgtx_stamp.so): its kernels stamp s_memtime /
s_memrealtime around the K loop and add the differences to a buffer of their own.
Usage: python tools/clock_probe.py [batch] [seconds per layer]      -> profiles/rNN_conv_clock.txt"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAMP = os.environ.get("GTX_STAMP_LIB") or os.path.join(ROOT, "geo-trax_amd", "build", "libgtx_stamp.so")
os.environ["GTX_LIB"] = STAMP
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))
from geotrax_amd import _lib, ops  # noqa: E402

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ctx = _lib.default_context(0)
dbg = ctypes.CDLL(STAMP)
for fn in (dbg.gtx_debug_conv_clock,):
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    fn.restype = ctypes.c_int


def read():
    """Sums of the two stamped translation units (a layer runs in one of them: the other's are zero)."""
    tot = [0] * 8
    for fn in (dbg.gtx_debug_conv_clock,):
        out = (ctypes.c_ulonglong * 8)()
        assert fn(out) == 0
        tot = [a + b for a, b in zip(tot, out)]
    return tot


LAYERS = [("h0.s1", 128, 192, 3, 1, 240), ("h0.s2c", 128, 128, 3, 1, 240), ("m4.m", 64, 64, 3, 1, 240), ("m6.m", 128, 128, 3, 1, 120),
          ("m8.m", 256, 256, 3, 1, 60), ("m5", 128, 256, 3, 2, 240), ("m4.cv2", 256, 128, 1, 1, 240), ("m9.cv2", 1024, 512, 1, 1, 60)]
print(f"# batch {NB}, {SECS:.0f} s of back-to-back launches per layer on random data (GTX_TIME_ZEROS=1: on zeros); clock = d(s_memtime) / d(s_memrealtime) x 100 MHz")
for nm, cin, cout, k, s, h in LAYERS:
    ops.conv2d_time(2, NB, h, h, cin, cout, k, s, iters=20, ctx=ctx)
    read()
    t0 = time.time()
    ms = fl = 0.0
    while time.time() - t0 < SECS:
        ms, fl = ops.conv2d_time(2, NB, h, h, cin, cout, k, s, iters=200, ctx=ctx)
    c, r, n, *ph = read()
    ghz = c / max(r, 1) * 0.1
    print(f"{nm:8s} {cin:4d}->{cout:4d} k{k} s{s} {h:4d}^2  {ms * 1000:8.1f} us  {fl / ms / 1e9:7.1f} TF/s   in-kernel clock {ghz:5.2f} GHz"
          f"   ({n} K loops stamped, {c / max(n, 1):9.0f} cycles per K loop; entry -> loop {ph[0] / max(n, 1):7.0f}, loop end -> stores retired {ph[1] / max(n, 1):7.0f})")
