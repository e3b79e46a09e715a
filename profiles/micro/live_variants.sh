#!/bin/bash
# times scratch builds of the library that differ in decode_live.hip's compile-time switches (ARPA workload, three calls in flight)
# usage (GPU box): bash profiles/micro/live_variants.sh <out dir under gpurun_out> "<flags of variant 1>" "<flags of variant 2>" ...
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-live_variants}; shift
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
B="python bench.py --workload arpa --no-cpu-baseline --no-side-figures --steps 30 --warmup 4"
i=0
for flags in "$@"; do
  i=$((i+1))
  rm -rf /tmp/rsvar && mkdir -p /tmp/rsvar && cp -r rhasspy_speech_amd include /tmp/rsvar/
  rm -f /tmp/rsvar/rhasspy_speech_amd/csrc/decode_live.o
  make -C /tmp/rsvar/rhasspy_speech_amd/csrc -j16 EXTRA="$flags" > $OUT/make_$i.log 2>&1 || { echo "variant $i ($flags): build failed"; tail -5 $OUT/make_$i.log; continue; }
  cp /tmp/rsvar/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  timeout 300 $B > $OUT/v$i.json 2> $OUT/v$i.err
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d=json.loads(open("$OUT/v$i.json").read().strip().splitlines()[-1])
    print("variant $i [$flags]: ms/step", round(d["ms_per_step"],2), "decode_ms", round(d.get("stages_ms",{}).get("decode",-1),2))
except Exception as e:
    print("variant $i [$flags] FAILED", e)
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
