#!/bin/bash
# Headline step with the samples in host memory (rs_decode_batch) against the same step with the samples resident in HBM
# (rs_decode_batch_device): per-call stage times and completion intervals of both, from one bench.py run with RS_BENCH_TRACE=1.
# usage (GPU box): bash profiles/micro/host_vs_device_input.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-host_vs_device}
mkdir -p $OUT
RS_BENCH_TRACE=1 timeout 600 python bench.py --no-cpu-baseline --steps 200 --warmup 8 > $OUT/line.json 2> $OUT/trace.txt
python - <<PY
import re, json
runs, cur = [], []
for ln in open("$OUT/trace.txt"):
    m = re.match(r"step (\d+) done at ([\d.]+) ms  timings \[(.*)\]", ln)
    if not m: continue
    k = int(m.group(1))
    if k == 0 and cur: runs.append(cur); cur = []
    cur.append((float(m.group(2)), [float(x) for x in m.group(3).split(",")]))
if cur: runs.append(cur)
for r in runs:
    if len(r) < 100: continue
    n = len(r); tm = [sum(x[1][j] for x in r[20:]) / (n - 20) for j in range(7)]
    print("steps %d  ms/step %.3f (steps 20..: %.3f)  mean timings h2d %.2f mfcc %.2f ivec %.2f nnet %.2f search %.2f d2h %.2f total %.2f" % (n, r[-1][0] / n, (r[-1][0] - r[19][0]) / (n - 20), *tm))
d = json.loads(open("$OUT/line.json").read().strip().splitlines()[-1])
print("headline", round(d["ms_per_step"], 3), "hbm_resident", round(d["hbm_resident"]["ms_per_step"], 3))
PY
