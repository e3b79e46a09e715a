cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r6g
for mr in 4 5 0; do
  RS_GEMM_B3J_MR=$mr timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6g/kt$mr -- python bench.py --steps 20 --warmup 2 --inflight 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
  f=$(find gpurun_out/r6g/kt$mr -name "*kernel_stats.csv" | head -1)
  echo "== MR $mr"; grep -E "GemmKernelB3" $f | cut -d, -f1-4 | cut -c 1-160
done
