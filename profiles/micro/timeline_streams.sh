# kernel trace of the streams workload (64 x 30 s, three queues) -> which kernel classes share the device
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-timeline_streams}
mkdir -p $OUT
python bench.py --workload streams --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python bench.py --workload streams --steps 16 --warmup 3 --no-cpu-baseline --no-side-figures > $OUT/line.json 2> $OUT/kt.log
TIMELINE_STREAMS=1 python profiles/micro/timeline.py $OUT/kt > $OUT/timeline_streams.txt 2>&1
cat $OUT/timeline_streams.txt
rm -rf $OUT/kt
