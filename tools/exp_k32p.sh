set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv2d" 2>&1 | tail -5
for v in 0 1 2; do
  echo "== GTX_K32P=$v rt_probe 1920"
  GTX_K32P=$v timeout 300 python tools/rt_probe.py 1920 1 1 2>&1 | grep -A3 "profile nb=2" | cut -c1-120
done
F="--no-cpu-baseline --no-f16-line --no-live-traffic"
for rep in 1 2; do
for v in 0 2; do
  for m in yolov8s rtdetr-l; do
    st=200; [ $m = rtdetr-l ] && st=60
    r=$(GTX_K32P=$v timeout 300 python bench.py --model $m --steps $st --warmup 10 $F < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
    echo "GTX_K32P=$v $m extract: $r"
  done
done
done
GTX_K32P=2 timeout 300 python tools/op_profile.py 2>&1 | tail -9 | cut -c1-140
GTX_K32P=2 GTX_PROFILE_PER_OP=1 timeout 300 python tools/rt_probe.py 1920 1 1 2>&1 | tail -5
