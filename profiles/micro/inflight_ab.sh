#!/bin/bash
# headline step against the number of decode calls in flight (RS_CONTEXTS decode contexts per model)
cd "$GRAFT_REPO_ROOT"
for cfg in "4 4" "8 4" "8 5" "8 6" "8 8" "4 4"; do
  set -- $cfg
  RS_CONTEXTS=$1 python bench.py --no-cpu-baseline --no-side-figures --steps 600 --warmup 20 --inflight $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $1 inflight $2: ms/step', round(d['ms_per_step'],3))"
done
RS_CONTEXTS=8 python bench.py --no-cpu-baseline --no-side-figures --steps 20 --warmup 5 --inflight 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver style, contexts 8 inflight 6: ms/step', round(d['ms_per_step'],3))"
python bench.py --no-cpu-baseline --no-side-figures --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver style, default: ms/step', round(d['ms_per_step'],3))"
