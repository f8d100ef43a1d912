set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rtdetr_gpu.py -q -x 2>&1 | tail -4
timeout 300 python tools/rt_probe.py 1920 1 1 2>&1 | grep -A12 "profile nb=2" | cut -c1-120
F="--no-cpu-baseline --no-f16-line --no-live-traffic --model rtdetr-l"
for rep in 1 2 3; do
  r=$(timeout 300 python bench.py --steps 60 --warmup 10 $F < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline'].get('token_linear',{}).get('avg_launch_us'))")
  echo "rtdetr-l extract: $r"
done
