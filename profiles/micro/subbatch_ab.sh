cd $GRAFT_REPO_ROOT
for sb in 2 1 2 1; do
  RS_SUBBATCHES=$sb python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('subbatches', $sb, 'ms_per_step', round(d['ms_per_step'],4), d['stages_ms'])"
done
for inf in 3 5 6; do
  python bench.py --steps 300 --warmup 20 --inflight $inf --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $inf, 'ms_per_step', round(d['ms_per_step'],4))"
done
