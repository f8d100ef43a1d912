"""How many kernels run at once in the timed region of a bench run: reads the kernel trace rocprofv3 wrote for
`rocprofv3 --kernel-trace --stats ... -- python3 bench.py` (tools/collect_profiles.sh: gpurun_out/final/stats) and prints, for the
densest 100 ms of convolution launches, the share of time with 0 / 1 / 2 / ... kernels in flight, the same for the convolution
and stem kernels alone, and how busy each hardware queue is. Usage: python tools/timeline.py DIR > profiles/rNN_timeline.txt"""
import collections
import csv
import glob
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/final/stats"
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
def is_conv(n):
    """Every convolution family of the detector: conv_igemm[_split], conv_k32_split (the dominant one since round 5 -- the round-5
    digest's filter missed it and reported 62 % where this prints ~90 %), conv_front_split, conv_wino*, the stems."""
    return "conv_" in n or "stem" in n


starts = [e[0] for e in ev if is_conv(e[2])]
W, best, j = 100e6, (0, None), 0
for i, s in enumerate(starts):
    while starts[j] < s - W:
        j += 1
    if i - j > best[0]:
        best = (i - j, (starts[j], s))
a, b = best[1]


def levels(pred, cap):
    pts = []
    for s, e, n, q in ev:
        if pred(n) and e > a and s < b:
            pts += [(max(s, a), 1), (min(e, b), -1)]
    pts.sort()
    hist, last, lvl = collections.Counter(), a, 0
    for t, dlt in pts:
        hist[min(lvl, cap)] += t - last
        last, lvl = t, lvl + dlt
    hist[min(lvl, cap)] += b - last
    return {k: round(v / (b - a), 3) for k, v in sorted(hist.items())}


print(f"# {f}: densest {(b - a) / 1e6:.0f} ms window ({best[0]} convolution launches)")
print("kernels in flight (share of time):          ", levels(lambda n: True, 6))
print("convolution / stem kernels in flight:       ", levels(is_conv, 4))
print("stabilizer / GMC kernels in flight:         ", levels(lambda n: any(k in n for k in ("pyr_", "fast_detect", "harris", "select_kernel", "describe", "match_kernel", "ransac", "gmc", "lk_")), 4))
for q in sorted({e[3] for e in ev}):
    iv = [(max(s, a), min(e, b)) for s, e, n, qq in ev if qq == q and e > a and s < b]
    print(f"hardware queue {q}: busy {sum(e - s for s, e in iv) / (b - a):.3f} of the window, {len(iv)} launches")
