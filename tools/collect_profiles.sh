#!/bin/bash
# collects everything profiles/ needs into gpurun_out/final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 60 --warmup 12 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $CMD > /dev/null 2> $O/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $CMD > /dev/null 2> $O/write.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $CMD > /dev/null 2> $O/sq.log
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.log
python bench.py --tracker botsort --no-cpu-baseline > $O/bench_botsort.json 2>/dev/null
python bench.py --half 0 --no-cpu-baseline --steps 60 > $O/bench_fp32.json 2>/dev/null
python bench.py --workload detect --batch 1 --det-streams 1 --no-cpu-baseline > $O/bench_detect_b1.json 2>/dev/null
python bench.py --workload detect --no-cpu-baseline > $O/bench_detect_2x2.json 2>/dev/null
python bench.py --workload register > $O/bench_register.json 2>/dev/null
python bench.py --workload georef > $O/bench_georef.json 2>/dev/null
python tools/conv_sweep.py 1920 2 > $O/sweep_b2.txt 2>/dev/null
python tools/conv_sweep.py 1920 4 > $O/sweep_b4.txt 2>/dev/null
ls $O
