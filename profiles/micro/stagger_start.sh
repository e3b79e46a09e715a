# the driver's invocation (--steps 20 --warmup 5) with the first calls of the timed region started RS_BENCH_STAGGER_MS apart
cd $GRAFT_REPO_ROOT
for st in ${STAGGERS:-0 0.3 0.6 0.9 1.2 0 0.6}; do
  RS_BENCH_STAGGER_MS=$st python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger', $st, 'ms_per_step', round(d['ms_per_step'],4), 'value', round(d['value']))"
done
