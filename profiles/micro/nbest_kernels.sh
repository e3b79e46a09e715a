#!/bin/bash
# per-kernel times of an n-best call (one call at a time): rocprofv3 kernel trace of profiles/micro/nbest_threads.py
# usage (GPU box): bash profiles/micro/nbest_kernels.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-nbest_kernels}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o nb -- python $GRAFT_REPO_ROOT/profiles/micro/nbest_threads.py > $OUT/prof.log 2>&1
python - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/prof/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for r in db.execute("select name,total_calls,average,percentage from top_kernels limit 10"):
        print("%-90s calls %5d avg_us %9.1f pct %.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3]))
PY
