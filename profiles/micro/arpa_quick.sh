#!/bin/bash
# ARPA workload, quick look: the config-2 parity tests, then step time and per-kernel times (one call in flight for the kernel times).
# usage (GPU box): bash profiles/micro/arpa_quick.sh <out dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-arpa_quick}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "config2 or arpa" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
B="python bench.py --workload arpa --no-cpu-baseline --no-side-figures"
for i in 1 2; do
timeout 300 $B --steps 30 --warmup 6 > $OUT/line$i.json 2> $OUT/line$i.err
python - <<PY
import json
d=json.loads(open("$OUT/line$i.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],3))
PY
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B --steps 6 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt.log
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:6]:
    print("    %-72s calls %4d avg_us %9.1f" % (r["Name"][:72], int(r["Calls"]), float(r["AverageNs"]) / 1000))
PY
