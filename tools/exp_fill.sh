set -u
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-f16-line --no-live-traffic"
for rep in 1 2; do
  GTX_BENCH_STAMPS=1 GTX_ENGINE_PROF=1 timeout 300 python bench.py --steps 20 --warmup 5 $F < /dev/null 2>&1 | grep -v "^{" | grep "stamps\|engine host\|stage marks" 
done
