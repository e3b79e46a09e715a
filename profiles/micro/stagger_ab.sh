#!/bin/bash
# GemmKernelB3J: order of the half-height tiles of a launch (RS_GEMM_B3J_STAGGER = 0 none first / 1 half the slots first / 2 alternating groups)
# on the strip form: headline step and per-kernel time.  -DRS_TUNING build in a scratch copy.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-stagger_ab}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -r rhasspy_speech_amd include /tmp/rstune/ && cp -r profiles/micro /tmp/rstune/profiles/ && rm -f /tmp/rstune/rhasspy_speech_amd/csrc/*.o
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so; cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
B="python bench.py --no-cpu-baseline --no-side-figures"
for st in ${STS:-1 0 2 1}; do
  RS_GEMM_B3J_STAGGER=$st timeout 120 $B --steps 300 --warmup 20 2>/dev/null | tail -1 > $OUT/line_$st.json
  python - <<PY
import json
d=json.loads(open("$OUT/line_$st.json").read())
print("stagger=$st ms/step", round(d["ms_per_step"],3), "nnet", round(d["stages_ms"]["nnet"],3))
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
