#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the files committed under profiles/.

    python tools/pmc_summary.py --stats DIR --fetch DIR --write DIR [--sq DIR] --batch B --tag r01

* --stats: a `rocprofv3 --kernel-trace --stats` run of `python3 bench.py ...`  -> profiles/<tag>_kernel_stats.csv
* --fetch / --write: two separate `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` runs of the same
  command (TCC slots do not fit both in one pass). Counter values are KB per dispatch. On gfx950 FETCH_SIZE
  reports half of the bytes of wide (16 B per lane) reads (MI355X_MICROARCH.md, HBM section), so HBM bytes per
  launch = 2 * FETCH_SIZE + WRITE_SIZE. Written to profiles/<tag>_pmc_traffic.json, which bench.py reads for
  roofline.traffic.
* --sq: optional SQ pass (SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, ...) -> MFMA busy fraction per kernel.
"""
import argparse
import collections
import csv
import glob
import json
import re
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def family(name: str) -> str:
    """rocprof kernel name -> the family name bench.py / gtx_detector_profile use."""
    m = re.match(r"_ZN3gtx\d+(conv_igemm2?_kernel)I(.*?)EEvNS_9ConvGroupE", name)
    if m:
        args = m.group(2)
        parts = []
        if args.startswith("DF16_"):
            parts.append("_Float16"); args = args[5:]
        elif args.startswith("f"):
            parts.append("float"); args = args[1:]
        parts += re.findall(r"Li(\d+)E", args)
        return f"{m.group(1)}<{', '.join(parts)}>"
    m = re.match(r"_ZN3gtx(?:\d+_GLOBAL__N_1)?\d+([a-z0-9_]+_kernel)", name)
    if m:
        return m.group(1)
    m = re.search(r"([A-Za-z0-9_]+_kernel)", name)
    return m.group(1) if m else name


def one(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        raise SystemExit(f"nothing matches {pattern}")
    return sorted(files)[-1]


def counters(d, wanted):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(one(f"{d}/**/*counter_collection.csv"))):
        if r["Counter_Name"] in wanted:
            a = acc[family(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--sq")
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--command", default="python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-profile")
    a = ap.parse_args()
    out = ROOT / "profiles"
    out.mkdir(exist_ok=True)
    if a.stats:
        shutil.copy(one(f"{a.stats}/**/*kernel_stats.csv"), out / f"{a.tag}_kernel_stats.csv")
    if a.fetch and a.write:
        f, w = counters(a.fetch, {"FETCH_SIZE"}), counters(a.write, {"WRITE_SIZE"})
        kernels = {}
        for k in sorted(set(f) | set(w)):
            fn, fv = f[k]["FETCH_SIZE"] if k in f else (0, 0.0)
            wn, wv = w[k]["WRITE_SIZE"] if k in w else (0, 0.0)
            fetch_kb = fv / fn if fn else 0.0
            write_kb = wv / wn if wn else 0.0
            kernels[k] = dict(dispatches=int(max(fn, wn)), fetch_size_kb_per_launch=round(fetch_kb, 1),
                              write_size_kb_per_launch=round(write_kb, 1),
                              hbm_bytes_per_launch=round((2.0 * fetch_kb + write_kb) * 1024.0))
        rec = dict(batch=a.batch, command=a.command,
                   method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only; "
                          "KB per dispatch averaged over all dispatches of the kernel; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                          "(gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
                   kernels=kernels)
        if a.sq:
            s = counters(a.sq, {"SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16"})
            for k, c in s.items():
                if k in kernels and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"][1] > 0:
                    busy = c["SQ_BUSY_CYCLES"][1]
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                        kernels[k]["mfma_busy_over_sq_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / busy, 4)
        (out / f"{a.tag}_pmc_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
        print(json.dumps({k: v for k, v in list(kernels.items())[:6]}, indent=1))


if __name__ == "__main__":
    main()
