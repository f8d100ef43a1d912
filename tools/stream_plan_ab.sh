#!/bin/bash
# Reproduces profiles/r04_stream_plan.txt on an MI355X: bench.py fed from host memory (and resident) under chosen stream-creation orders,
# a fresh process per number. GTX_ENGINE_ORDER tokens: d detector, s stabilizer, f feeder copy stream, g GMC, n null stream, x idle stream.
# "d,n,f,s,s,s,s,d" is what the engine did before the stream plan (the null stream came into being inside the first Detector constructor):
# both detectors on one hardware queue. Usage: bash tools/stream_plan_ab.sh > gpurun_out/stream_plan_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(d.get("value"),1))'
C="--no-cpu-baseline --no-profile --no-f16-line --steps 150 --warmup 10"
run() {   # label, order ("" = the plan's own), extra bench flags
  r=""; h=""
  for rep in 1 2; do
    r="$r $(GTX_ENGINE_ORDER=$2 timeout 200 python $R/bench.py $C $3 2>/dev/null | python -c "$J")"
    h="$h $(GTX_ENGINE_ORDER=$2 timeout 200 python $R/bench.py --host-frames $C $3 2>/dev/null | python -c "$J")"
  done
  echo "$1 [${2:-plan}] $3: resident $r | host $h"
}
run "the plan (each detector a queue of its own)" ""
run "before the plan: both detectors on queue 1" d,n,f,s,s,s,s,d
run "feeder on a detector's queue" d,d,s,s,s,s,f,n,x,g
run "a detector on the stabilizers' queue" d,d,s,s,n,x,s,x,x,x,s,x,x,x,f,x,x,x,g
run "stabilizers on one queue, feeder on its own" d,d,f,s,s,n,x,x,s,x,x,x,s,x,x,x,g
run "three isolated detector streams" d,d,d,s,s,n,x,x,s,x,x,x,s,x,x,x,f,x,x,x,g "--det-streams 3"
for p in 1,0 0,1 0,-1; do GTX_ENGINE_PRIO=$p run "stream priorities $p" ""; done
