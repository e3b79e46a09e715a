cd $GRAFT_REPO_ROOT
for nt in 1024 512 256; do
RS_REG_NT=$nt python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$nt', d[\"value\"], d[\"ms_per_step\"], d[\"stages_ms\"][\"decode\"])"
done
