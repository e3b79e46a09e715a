cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02t
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02t/kt -- python bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r02t/line.json 2> gpurun_out/r02t/kt.log
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -E "passed|failed" > gpurun_out/r02t/tests.log
