#!/bin/bash
# RegDecodeKernel shape in the pipelined headline step: default (256 threads for crowded batches) against RS_REG_NT=512 (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-reg_nt_ab}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -r rhasspy_speech_amd include /tmp/rstune/ && cp -r profiles/micro /tmp/rstune/profiles/ && rm -f /tmp/rstune/rhasspy_speech_amd/csrc/*.o
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so; cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
B="python bench.py --no-cpu-baseline --no-side-figures"
for v in "" "RS_REG_NT=512" "" "RS_REG_NT=512"; do
  env $v timeout 120 $B --steps 400 --warmup 20 2>/dev/null | tail -1 > $OUT/line.json
  python - <<PY
import json
d=json.loads(open("$OUT/line.json").read())
print("[$v] ms/step", round(d["ms_per_step"],3), "search", round(d["stages_ms"]["decode"],3))
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
