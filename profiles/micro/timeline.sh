# kernel trace of the pipelined headline run (four calls in flight) -> which kernel classes share the device, per call
# usage (GPU box): bash profiles/micro/timeline.sh <out> [inflight]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-timeline}
INF=${2:-4}
mkdir -p $OUT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python bench.py --steps 60 --warmup 8 --inflight $INF --no-cpu-baseline --no-side-figures > $OUT/line.json 2> $OUT/kt.log
python profiles/micro/timeline.py $OUT/kt > $OUT/timeline_inflight$INF.txt 2>&1
cat $OUT/timeline_inflight$INF.txt
rm -rf $OUT/kt
