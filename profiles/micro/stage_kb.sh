cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for kb in 12 20 32 44; do
  for infl in 1 4; do
    RS_DECODE_STAGE_KB=$kb timeout 300 python bench.py --steps 200 --warmup 12 --inflight $infl --no-cpu-baseline --no-side-figures 2>/dev/null | tail -n 1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('stage $kb KB inflight $infl: %.3f ms/step' % l['ms_per_step'], {k: round(v, 2) for k, v in l['stages_ms'].items()})"
  done
done
