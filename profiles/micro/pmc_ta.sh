#!/bin/bash
# LDS pipe counters per kernel of the headline batch (one call in flight): is a kernel waiting for the CU's one LDS pipe?
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_lds}
mkdir -p $OUT
P="python bench.py --workload ${2:-grammar} --no-cpu-baseline --no-side-figures --steps ${3:-3} --warmup 1 --inflight 1"
timeout 400 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- $P > /dev/null 2> $OUT/pmc.log
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open("$f")):
    k = r["Kernel_Name"][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("TA_TA_BUSY", 0))[:14]:
    c = max(n[k], 1)
    print(k.ljust(48), "launches", c, {kk: round(vv / c / 1e6, 2) for kk, vv in v.items()}, "(millions per launch)")
PY
