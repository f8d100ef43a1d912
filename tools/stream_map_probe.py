"""Which hardware queue a HIP stream lands on, as a function of creation order and of first-use order.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/stream_map_probe.py SCENARIO
    python tools/stream_map_probe.py --read DIR
SCENARIO: comma-separated steps: cK = create stream K, tK = launch a marker kernel on stream K (+ synchronize), dK = destroy stream K,
n0 = gtx_device_open_null_stream (the null stream takes its place now).
Stream K's marker kernel is gtx_yuv420_to_bgr_dev on a 64 x (32 * (K + 1)) frame: the grid size identifies the stream in the trace."""
import collections
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "geo-trax_amd"))

if sys.argv[1] == "--read":
    f = sorted(glob.glob(sys.argv[2] + "/**/*_kernel_trace.csv", recursive=True))[-1]
    seen = collections.OrderedDict()
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
        if "yuv420" in r["Kernel_Name"]:
            seen.setdefault(int(r["Grid_Size_Y"]) // 16 - 1, []).append(r["Queue_Id"])
    print("  ".join(f"s{k}->q{','.join(sorted(set(v)))}" for k, v in seen.items()))
    sys.exit(0)

from geotrax_amd import _lib  # noqa: E402

streams = {}
W = 64
buf_in = buf_out = None
for step in sys.argv[1].split(","):
    op, k = step[0], int(step[1:])
    if op == "c":
        streams[k] = _lib.Context(0)
        if buf_in is None:
            buf_in, buf_out = streams[k].dev_alloc(W * 32 * 16 * 3), streams[k].dev_alloc(W * 32 * 16 * 3)
    elif op == "n":
        _lib.check(_lib.load().gtx_device_open_null_stream(0))
    elif op == "t":
        c = streams[k]
        _lib.check(c.lib.gtx_yuv420_to_bgr_dev(c.handle, C.c_void_p(buf_in), 32 * (k + 1), W, C.c_void_p(buf_out)))
        c.synchronize()
    elif op == "d":
        streams.pop(k).close()
print("done", sys.argv[1])
