#!/bin/bash
# quick look: parity tests named by $2 (pytest -k), then per-kernel times of the headline batch under rocprofv3 (one call in flight)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-kt_quick}
mkdir -p $OUT
if [ -n "$2" ]; then timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "$2" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log; fi
B="python bench.py --no-cpu-baseline --no-side-figures ${3:-}"
$B --steps 300 --warmup 20 > $OUT/line.json 2> $OUT/line.err
python - <<PY
import json
d=json.loads(open("$OUT/line.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],3), "stages", {k: round(v,3) for k,v in d.get("stages_ms",{}).items()})
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 20 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt.log
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
ref=max(int(r["Calls"]) for r in rows if "MfccKernel" in r["Name"] or "HashDecode" in r["Name"] or "LiveDecode" in r["Name"])
for r in rows[:22]:
    print("    %-72s per-call %5.1f avg_us %8.1f" % (r["Name"][:72], int(r["Calls"]) / ref, float(r["AverageNs"]) / 1000))
PY
