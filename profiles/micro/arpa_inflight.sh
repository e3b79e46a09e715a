#!/bin/bash
# ARPA workload: step time against the number of calls in flight and the number of hardware queues the HIP runtime multiplexes the
# library's streams onto (GPU_MAX_HW_QUEUES, default 4).  usage (GPU box): bash profiles/micro/arpa_inflight.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-arpa_inflight}
mkdir -p $OUT
B="python bench.py --workload arpa --no-cpu-baseline --no-side-figures --steps 30 --warmup 6"
run() { # name, inflight, env...
  name=$1; inf=$2; shift; shift
  env "$@" timeout 300 $B --inflight $inf > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/step", round(d["ms_per_step"],2))
except Exception as e:
    print("$name FAILED", e)
PY
}
run q4_if3 3 RS_CONTEXTS=8
run q4_if4 4 RS_CONTEXTS=8
run q8_if3 3 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=8
run q8_if4 4 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=8
run q8_if6 6 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=8
run q16_if4 4 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=16
run q16_if6 6 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=16
run q2_if3 3 RS_CONTEXTS=8 GPU_MAX_HW_QUEUES=2
