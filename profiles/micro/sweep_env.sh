# headline step under a list of environment settings: bash profiles/micro/sweep_env.sh <out> "VAR=val VAR2=val;--inflight N" ...
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-sweep}
shift
mkdir -p $OUT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
for spec in "$@"; do
  envs="${spec%%;*}"; args="${spec#*;}"
  line=$(env $envs python bench.py --steps 300 --warmup 12 --no-cpu-baseline --no-side-figures $args 2>/dev/null | tail -n 1)
  echo "$spec -> $(echo "$line" | python -c 'import json,sys; l=json.loads(sys.stdin.read()); print("%.3f ms/step, %d audio-s/s" % (l["ms_per_step"], l["value"]))')" | tee -a $OUT/sweep.txt
done
