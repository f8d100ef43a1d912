#!/bin/bash
# gpurun_out/final/ (tools/collect_profiles.sh, merged back by gpurun) -> profiles/<tag>_*: the files that are committed.
# Usage: tools/publish_profiles.sh r04
set -e
T=${1:?tag}; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/final; P=$R/profiles
python $R/tools/pmc_summary.py --stats $O/stats --fetch $O/fetch --write $O/write --sq $O/sq1,$O/sq2 --batch 2 --tag $T
for f in default botsort ocsort deepocsort fp32_exact f16 detect_b1 detect_2x2 register register_ortho15000 georef warp extract_georef cli host_frames rtdetr rtdetr_detect one_det_stream; do
  [ -s $O/bench_$f.json ] && cp $O/bench_$f.json $P/${T}_bench_$f.json
done
[ -s $O/bench_under_rocprof.json ] && cp $O/bench_under_rocprof.json $P/${T}_bench_under_rocprof.json
for f in op_profile_f32s_b2 op_profile_f16_b2 op_profile_f32_b2 op_profile_per_launch_f32s_b2 conv_layer_sweep_b2 conv_layer_sweep_b2_zeros conv_clock rtdetr_op_profile; do
  [ -s $O/$f.txt ] && cp $O/$f.txt $P/${T}_$f.txt
done
python $R/tools/timeline.py $O/stats > $P/${T}_timeline.txt
python - <<PY
import csv, glob, collections
# per-kernel stats of two more rocprofv3 runs: BoT-SORT bench, the stabilizer alone
for d, out in (("$O/stats_botsort", "$P/${T}_kernel_stats_botsort.csv"), ("$O/stats_iso", "$P/${T}_kernel_stats_isolated.csv"), ("$O/stats_stab", "$P/${T}_kernel_stats_stabilizer.csv"), ("$O/stats_rtdetr", "$P/${T}_kernel_stats_rtdetr.csv")):
    fs = sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True))
    if fs:
        open(out, "w").write(open(fs[-1]).read())
PY
ls $P | grep "^$T" | wc -l
