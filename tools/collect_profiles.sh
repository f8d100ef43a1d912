#!/bin/bash
# Collects everything profiles/rNN_* is made from into gpurun_out/final/ (run on an MI355X; then tools/pmc_summary.py
# and plain copies turn it into the committed files; delete the local gpurun_out/final first, gpurun only ever adds files). rocprofv3 counter passes are separate runs with --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-profile --no-f16-line"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-f16-line > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_botsort -- python3 $R/bench.py --tracker botsort --steps 60 --warmup 12 --no-cpu-baseline --no-f16-line --no-profile > /dev/null 2> $O/stats_botsort.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > /dev/null 2> $O/write.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq1 -- python3 $R/tools/op_profile.py 2 f32s 3 > $O/sq1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/op_profile.py 2 f32s 3 > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_iso -- python3 $R/tools/op_profile.py 2 f32s 5 > $O/op_profile_under_rocprof.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_stab -- python3 $R/tools/stab_profile.py 60 > $O/stab_profile.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_rtdetr -- python3 $R/bench.py --model rtdetr-l --steps 40 --warmup 6 --no-cpu-baseline --no-f16-line --no-live-traffic > $O/bench_rtdetr_under_rocprof.json 2> $O/stats_rtdetr.log
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log
python bench.py --tracker botsort --no-cpu-baseline --no-f16-line > $O/bench_botsort.json 2>/dev/null
python bench.py --tracker ocsort --no-cpu-baseline --no-f16-line > $O/bench_ocsort.json 2>/dev/null
python bench.py --tracker deepocsort --no-cpu-baseline --no-f16-line > $O/bench_deepocsort.json 2>/dev/null
python bench.py --fp32 exact --no-cpu-baseline --no-f16-line --steps 60 > $O/bench_fp32_exact.json 2>/dev/null
python bench.py --half 1 --no-cpu-baseline > $O/bench_f16.json 2>/dev/null
python bench.py --workload detect --batch 1 --det-streams 1 --no-cpu-baseline > $O/bench_detect_b1.json 2>/dev/null     # carries `pipelined` (2 / 3 passes in flight)
python bench.py --workload cli > $O/bench_cli.json 2>/dev/null                                                          # the product from a .y4m / .npy file
python bench.py --host-frames --no-cpu-baseline --no-f16-line --no-profile > $O/bench_host_frames.json 2>/dev/null
python bench.py --workload detect --no-cpu-baseline --no-f16-line > $O/bench_detect_2x2.json 2>/dev/null
python bench.py --workload register > $O/bench_register.json 2>/dev/null
python bench.py --workload register --ortho 15000 --steps 3 > $O/bench_register_ortho15000.json 2>/dev/null   # K11 at the reference's size
python bench.py --workload georef > $O/bench_georef.json 2>/dev/null
python bench.py --workload warp > $O/bench_warp.json 2>/dev/null
python bench.py --workload extract+georef > $O/bench_extract_georef.json 2>/dev/null
python bench.py --model rtdetr-l --steps 100 < /dev/null > $O/bench_rtdetr.json 2>/dev/null                  # the reference's RTDETR branch
python bench.py --model rtdetr-l --workload detect --no-cpu-baseline --steps 100 < /dev/null > $O/bench_rtdetr_detect.json 2>/dev/null
python bench.py --det-streams 1 --no-cpu-baseline --no-f16-line --steps 200 < /dev/null > $O/bench_one_det_stream.json 2>/dev/null   # the front launch in situ with one detector stream
timeout 300 python tools/rt_probe.py 1920 1 1 < /dev/null > $O/rtdetr_op_profile.txt 2>&1
for p in f32s f16 f32; do python tools/op_profile.py 2 $p 10 > $O/op_profile_${p}_b2.txt 2>/dev/null; done
python tools/conv_sweep.py 1920 2 all f32s f16 > $O/conv_layer_sweep_b2.txt 2>/dev/null
GTX_TIME_ZEROS=1 python tools/conv_sweep.py 1920 2 all f32s > $O/conv_layer_sweep_b2_zeros.txt 2>/dev/null
GTX_PROFILE_PER_OP=1 python tools/op_profile.py 2 f32s 10 > $O/op_profile_per_launch_f32s_b2.txt 2>/dev/null
python tools/clock_probe.py 2 2 > $O/conv_clock.txt 2>/dev/null
GTX_TIME_ZEROS=1 python tools/clock_probe.py 2 2 > $O/conv_clock_zeros.txt 2>/dev/null
ls $O
