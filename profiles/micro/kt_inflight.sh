#!/bin/bash
# per-kernel times of the headline batch with the default four calls in flight (kernels overlap: durations include sharing the device)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-kt_inflight}
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python bench.py --no-cpu-baseline --no-side-figures --steps 100 --warmup 10 > $OUT/line.json 2> $OUT/kt.log
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$f")))
for r in rows[:16]: print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1000, 1), r["Percentage"])
print("ms/step", round(json.loads(open("$OUT/line.json").read().strip().splitlines()[-1])["ms_per_step"], 3))
PY
