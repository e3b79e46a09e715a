#!/bin/bash
# Hardware counters of the large-graph search kernel on the ARPA workload, one rocprofv3 pass per counter group (counters only:
# no trace domains).  usage (GPU box): bash profiles/micro/pmc_search.sh <out dir under gpurun_out> [env assignments ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pmc_search}; shift
mkdir -p $OUT
B="python bench.py --workload arpa --no-cpu-baseline --no-side-figures --steps 1 --warmup 1 --inflight 1"
env "$@" $B > /dev/null 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $B > /dev/null 2> $OUT/p$i.log || echo "pass $i ($grp) failed: $(tail -2 $OUT/p$i.log)"
done <<GROUPS
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum
TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_READ_sum TCC_WRITE_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WAVEFRONTS_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_UTCL2_BUSY
GROUPS
python - <<PY
import csv, glob, collections
tot = collections.OrderedDict()
for d in sorted(glob.glob("$OUT/p*/"), key=lambda s: int(s.rstrip('/').split('p')[-1])):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "DecodeKernel" not in r["Kernel_Name"] or "Lattice" in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void rs::", "").replace("(anonymous namespace)::", "")
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        # a launch has one row per (dispatch, counter): mean per dispatch, over the dispatches that did work (the fallback kernel exits at once)
        for (k, c), (v, n) in acc.items():
            tot[(k, c)] = v / max(n, 1)
with open("$OUT/summary.txt", "w") as fh:
    for (k, c), v in tot.items():
        line = f"{k:60s} {c:40s} {v:18.0f}"
        print(line); fh.write(line + "\n")
PY
