cd "$GRAFT_REPO_ROOT"
python bench.py --workload streams --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
for e in "A=1" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=4" "RS_CONTEXTS=1" "RS_CONTEXTS=1 GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=2"; do
  env $e timeout 300 python bench.py --workload streams --steps 20 --warmup 3 --no-cpu-baseline --no-side-figures 2>/dev/null | tail -n 1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$e: %.2f ms/step' % l['ms_per_step'])"
done
