# wave priority of RegDecodeKernel (s_setprio) against the headline step: 3 (compiled in) vs 0..2
cd $GRAFT_REPO_ROOT
for p in 3 0 1 2 3 0; do
  RS_REG_PRIO_RT=$p python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio', $p, 'ms_per_step', round(d['ms_per_step'],4), 'decode stage', round(d['stages_ms']['decode'],3), 'nnet', round(d['stages_ms']['nnet'],3))"
done
