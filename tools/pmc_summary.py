#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the files committed under profiles/.

    python tools/pmc_summary.py --stats DIR --fetch DIR --write DIR [--sq DIR] --batch B --tag r01

* --stats: a `rocprofv3 --kernel-trace --stats` run of `python3 bench.py ...`  -> profiles/<tag>_kernel_stats.csv
* --fetch / --write: two separate `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` runs of the same
  command (TCC slots do not fit both in one pass). Counter values are KB per dispatch. On gfx950 FETCH_SIZE
  reports half of the bytes of wide (16 B per lane) reads (MI355X_MICROARCH.md, HBM section), so HBM bytes per
  launch = 2 * FETCH_SIZE + WRITE_SIZE. Written to profiles/<tag>_pmc_traffic.json, which bench.py reads for
  roofline.traffic.
* --sq DIR[,DIR]: SQ / GRBM passes of tools/op_profile.py (launches alone on their stream) -> profiles/<tag>_pmc_sq.json:
  normalised matrix-pipe utilisation, LDS activity and bank conflicts, wave stall shares, occupancy per kernel family.
"""
import argparse
import collections
import csv
import glob
import json
import re
import os
import shutil
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def family(name: str) -> str:
    """rocprof kernel name -> the family name bench.py / gtx_detector_profile use."""
    m = re.search(r"conv_igemm_split_kernel<([^>]*)>", name)             # demangled (anonymous namespace) form
    if m:
        return f"conv_igemm_split_kernel<{m.group(1)}>"
    m = re.match(r"_ZN3gtx12_GLOBAL__N_123conv_igemm_split_kernelI(.*?)EEvNS_9ConvGroupE", name)
    if m:
        return "conv_igemm_split_kernel<" + ", ".join(re.findall(r"Li(\d+)E", m.group(1))) + ">"
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
    return max(files, key=os.path.getmtime)     # gpurun merges new files into the directory: older collections may still be there


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
    ap.add_argument("--command", default="python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-profile --no-f16-line")
    ap.add_argument("--sq-command", default="python3 tools/op_profile.py 2 f32s 3")
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
        (out / f"{a.tag}_pmc_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
        print(json.dumps({k: v for k, v in list(kernels.items())[:6]}, indent=1))
    if a.sq:
        sq_summary(a.sq.split(","), out / f"{a.tag}_pmc_sq.json", a.sq_command)


N_SIMD, N_CU, N_XCD = 1024, 256, 8


def sq_summary(dirs, dst, command):
    """Normalised SQ / GRBM counters per kernel family from one or more `rocprofv3 --pmc ... --kernel-trace` passes of
    tools/op_profile.py (every launch alone on its stream). Per launch averages; units per MI355X_MICROARCH.md:
    SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_f16), SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
    count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs.
      mfma_busy_pct      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs)      the matrix-pipe utilisation
      lds_active_pct     = SQ_LDS_IDX_ACTIVE / (kernel cycles * 256 CUs)
      lds_conflict_pct   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
      wave_*_pct         = shares of SQ_WAVE_CYCLES (waiting in s_waitcnt/barrier, issue-stalled, issuing)
      occupancy_waves_per_simd = SQ_WAVE_CYCLES * 4 / (kernel cycles * 1024)"""
    want = {"SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
            "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD"}
    acc = collections.defaultdict(dict)
    for d in dirs:
        for k, c in counters(d, want).items():
            for name, (n, v) in c.items():
                acc[k][name] = v / n
    out = {}
    for k, c in acc.items():
        if "GRBM_GUI_ACTIVE" not in c or "SQ_WAVE_CYCLES" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / N_XCD
        w = c["SQ_WAVE_CYCLES"]
        r = {"kernel_cycles": round(cyc), "mfma_instructions": round(c.get("SQ_INSTS_MFMA", 0)),
             "mfma_busy_pct": round(100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * N_SIMD), 1),
             "lds_active_pct": round(100 * c.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * N_CU), 1),
             "lds_conflict_pct": round(100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1), 1),
             "wave_waiting_pct": round(100 * c.get("SQ_WAIT_ANY", 0) / w, 1), "wave_issue_stalled_pct": round(100 * c.get("SQ_WAIT_INST_ANY", 0) / w, 1),
             "wave_issuing_pct": round(100 * c.get("SQ_ACTIVE_INST_ANY", 0) / w, 1), "wave_lds_issue_stall_pct": round(100 * c.get("SQ_WAIT_INST_LDS", 0) / w, 1),
             "occupancy_waves_per_simd": round(4 * w / (cyc * N_SIMD), 2), "valu_per_mfma": round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_INSTS_MFMA", 0), 1), 1)}
        out[k] = r
    dst.write_text(json.dumps({"command": command, "method": sq_summary.__doc__, "kernels": out}, indent=1) + "\n")
    for k in list(out)[:6]:
        print(k, out[k])


if __name__ == "__main__":
    main()
