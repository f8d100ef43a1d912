#!/bin/bash
# usage: ab2.sh N libA libB ... : detector-only, 1 stream and 2 streams, then full extract
N=$1; shift
run() { GTX_LIB=$1 python bench.py --no-cpu-baseline --no-profile --steps $3 $2 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f' % d['value'])"; }
for i in $(seq $N); do
  for lib in "$@"; do
    a=$(run $lib "--workload detect --det-streams 1" 150)
    b=$(run $lib "--workload detect --det-streams 2" 150)
    c=$(run $lib "" 300)
    echo "$(basename $lib) det1 $a det2 $b extract $c"
  done
done
