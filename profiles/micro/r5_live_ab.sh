#!/bin/bash
# first GPU run of round 5's live-state-table search: parity tests, then A/B timing against round 4's kernel
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r5_live_ab
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "variants_agree or live_state_table or lattice" > $OUT/pytest_parity.log 2>&1
echo "parity rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_parity.log
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -k "config2" > $OUT/pytest_c2.log 2>&1
echo "config2 rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_c2.log
B="python bench.py --workload arpa --no-cpu-baseline --no-side-figures"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 $B --steps 30 --warmup 4 > $OUT/$name.json 2> $OUT/$name.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d=json.loads(open("$OUT/$name.json").read().strip().splitlines()[-1])
    print("$name", "ms/step", round(d["ms_per_step"],2), "decode_ms", round(d.get("stages_ms",{}).get("decode",-1),2), "checked:", d.get("results_checked","")[:60])
except Exception as e:
    print("$name FAILED", e)
PY
}
run r4_if3 RS_DECODER=hash_r4
for sh in ${SHAPES:-1024 512 5122 2564}; do run live${sh}_if3 RS_LIVE_SHAPE=$sh; done
B="$B --inflight 4"
for sh in ${SHAPES:-1024 512 5122 2564}; do run live${sh}_if4 RS_LIVE_SHAPE=$sh; done
