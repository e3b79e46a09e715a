#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against.  Run on the GPU box:
#   gpurun -- 'bash profiles/collect.sh r01'
# Writes gpurun_out/<tag>/...; copy the summaries into profiles/<tag>/ (see profiles/README.md).
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1      # page the image in
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.log
# (bench.py measures stage times and the roofline on un-overlapped calls, so the kernel trace and the counters are taken with one call
# in flight; the headline line at the end uses the default, four)
# PMC passes, each on its own (no trace domains besides the counters)
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/pmc_fetch.log
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/pmc_write.log
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/pmc_sq.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.log                       # the default invocation, as the driver runs it
python bench.py --no-cpu-baseline --prune-output > $OUT/bench_line_pruned_output.json 2>> $OUT/bench.log
python bench.py --no-cpu-baseline --inflight 1 > $OUT/bench_line_one_call_in_flight.json 2>> $OUT/bench.log
find $OUT -name "*.csv" | head -20
