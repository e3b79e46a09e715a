cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02j
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02j/kt_img -- python bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r02j/img.json 2> gpurun_out/r02j/img.log
RS_GEMM_B3I=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02j/kt_noimg -- python bench.py --steps 10 --warmup 2 --inflight 1 --no-cpu-baseline > gpurun_out/r02j/noimg.json 2> gpurun_out/r02j/noimg.log
find gpurun_out/r02j -name "*kernel_stats.csv"
