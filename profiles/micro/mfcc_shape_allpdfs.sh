#!/bin/bash
# MfccKernel shape (RS_MFCC_SHAPE 0 / 16) on the all-pdfs model of the headline workload (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfcc_shape_allpdfs}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for sh in 16 0 16 0; do
  for nf in 3 4 5; do
    RS_CONTEXTS=8 RS_MFCC_SHAPE=$sh python bench.py --all-pdfs --inflight $nf --no-cpu-baseline --no-side-figures --steps 300 --warmup 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape $sh all-pdfs inflight $nf: ms/step', round(d['ms_per_step'],3))"
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
