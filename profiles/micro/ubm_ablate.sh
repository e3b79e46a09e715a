cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02v
for a in 0 1; do
RS_UBM_ABLATE=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02v/kt$a -- python bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r02v/kt$a.log
done
