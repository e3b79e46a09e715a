#!/bin/bash
# where a streams step (configs[4], 64 x 30 s) spends the host's time and the queues' time: -DRS_TUNING build, RS_STREAMS_TRACE=1
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-streams_trace}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -r rhasspy_speech_amd include /tmp/rstune/ && cp -r profiles/micro /tmp/rstune/profiles/ && rm -f /tmp/rstune/rhasspy_speech_amd/csrc/*.o
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j16 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so; cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
RS_STREAMS_TRACE=1 RS_BENCH_TRACE=1 python bench.py --workload streams --no-cpu-baseline --no-side-figures --steps 10 --warmup 2 > $OUT/line.json 2> $OUT/trace.err
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
grep -i "streams\|advance\|accept\|finish\|host" $OUT/trace.err | tail -25
python - <<PY
import json
d=json.loads(open("$OUT/line.json").read().strip().splitlines()[-1]); print("ms/step", round(d["ms_per_step"],2), d.get("stages_ms"))
PY
