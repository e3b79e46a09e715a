#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash profiles/collect.sh r02 grammar'        (workload: grammar | arpa | streams | mixed)
# Writes gpurun_out/<tag>_<workload>/...; profiles/summarize.py condenses that into profiles/<round>/<workload>_*.
set -e
TAG=${1:-r02}
WL=${2:-grammar}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_${WL}
mkdir -p $OUT
case $WL in
  grammar) KT_STEPS=20; PMC_STEPS=3;;
  arpa)    KT_STEPS=6;  PMC_STEPS=2;;
  streams) KT_STEPS=2;  PMC_STEPS=1;;
  mixed)   KT_STEPS=6;  PMC_STEPS=2;;
esac
B="python bench.py --workload $WL --no-cpu-baseline"
P="$B --no-side-figures"      # under the profiler: the headline steps and the un-overlapped stage calls only
$B --steps 2 --warmup 1 > /dev/null 2>&1      # page the image in
# (bench.py measures stage times and the roofline on un-overlapped calls, so the kernel trace and the counters are taken with one call
# in flight; the headline line at the end uses the workload's default)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $P --steps $KT_STEPS --warmup 2 --inflight 1 > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.log
# PMC passes, each on its own (no trace domains besides the counters)
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $P --steps $PMC_STEPS --warmup 1 --inflight 1 > /dev/null 2> $OUT/pmc_fetch.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $P --steps $PMC_STEPS --warmup 1 --inflight 1 > /dev/null 2> $OUT/pmc_write.log
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $P --steps $PMC_STEPS --warmup 1 --inflight 1 > /dev/null 2> $OUT/pmc_sq.log
python bench.py --workload $WL > $OUT/bench_line.json 2> $OUT/bench.log                       # the default invocation, as the driver runs it
if [ "$WL" = grammar ]; then
  $B --all-pdfs > $OUT/bench_line_all_pdfs.json 2>> $OUT/bench.log
  $B --inflight 1 > $OUT/bench_line_one_call_in_flight.json 2>> $OUT/bench.log
fi
find $OUT -name "*.csv" | head -20
