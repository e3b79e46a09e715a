# RS_SEARCH_CUS: the grammar-graph search confined to part of the CUs (hipExtStreamCreateWithCUMask), headline step
cd $GRAFT_REPO_ROOT
for n in 0 128 160 192 224 0 128; do
  RS_SEARCH_CUS=$n python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('search CUs', $n, 'ms_per_step', round(d['ms_per_step'],4), 'decode stage', round(d['stages_ms']['decode'],3), 'call', round(d['stages_ms']['total_call'],3))"
done
