cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ubm
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
for a in 0 1; do
RS_BENCH_DEBUG_UNCHECKED=1 RS_UBM_ABLATE=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ubm/kt$a -- python bench.py --steps 5 --warmup 2 --inflight 1 --no-cpu-baseline --no-side-figures > /dev/null 2> gpurun_out/ubm/kt$a.log
echo "RS_UBM_ABLATE=$a: $(python profiles/micro/kt_top.py gpurun_out/ubm/kt$a 12 | grep -i ubm)"
done
rm -rf gpurun_out/ubm
