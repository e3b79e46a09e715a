#!/bin/bash
# Driver-style runs (20 steps from a standing start, 5 warm-up steps) of the headline workload against the number of calls in flight.
# usage (GPU box): bash profiles/micro/inflight20.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-inflight20}
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-side-figures"
for rep in 1 2 3 4 5; do
  for inf in 2 3 4 5 6; do
    timeout 200 $B --steps 20 --warmup 5 --inflight $inf > $OUT/s20_i${inf}_$rep.json 2> /dev/null
  done
done
python - <<PY
import json, glob
for inf in (2, 3, 4, 5, 6):
    v = []
    for f in sorted(glob.glob("$OUT/s20_i%d_*.json" % inf)):
        try: v.append(round(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"], 3))
        except Exception as e: v.append(None)
    print("inflight", inf, v, "mean", round(sum(x for x in v if x) / max(1, len([x for x in v if x])), 3))
PY
