#!/bin/bash
# IvecAccumKernel's LDS footprint beside other calls' kernels: frames per chunk 320 (113 KB of LDS: only CUs without a layer-GEMM workgroup
# can take it) against 160 (75 KB), scratch builds with -DRS_ACC_TC; kernel time alone and in flight, pipelined step
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-acc_tc}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for tc in 320 160 320 160; do
  rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
  rm -f /tmp/rstune/rhasspy_speech_amd/csrc/ivector_kernels.o
  make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA="-DRS_ACC_TC=$tc" > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
  cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  python bench.py --no-cpu-baseline --no-side-figures --steps 400 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TC $tc: ms/step', round(d['ms_per_step'],3), 'ivector stage alone', round(d['stages_ms']['ivector'],3))"
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
