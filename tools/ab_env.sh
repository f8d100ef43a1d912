#!/bin/bash
# usage: ab_env.sh N "VAR=a" "VAR=b" ... : the driver's bench line (value) and detector-only under each environment setting,
# interleaved N times on one box (boxes of the pool differ by up to 8 %: only same-box interleaved numbers compare)
N=$1; shift
run() { env $1 python bench.py --no-cpu-baseline --no-profile --no-live-traffic --no-f16-line --steps $3 $2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f' % d['value'])"; }
for i in $(seq $N); do
  for e in "$@"; do
    b=$(run "$e" "--workload detect --det-streams 2" 200)
    c=$(run "$e" "" 300)
    echo "$e det2 $b extract $c"
  done
done
