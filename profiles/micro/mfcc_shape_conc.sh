#!/bin/bash
# MfccKernel shape (RS_MFCC_SHAPE 0 / 16) against the number of calls in flight: headline batch and the mixed workload (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfcc_shape_conc}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
run() { RS_CONTEXTS=8 RS_MFCC_SHAPE=$1 python bench.py --workload $2 --inflight $3 --no-cpu-baseline --no-side-figures --steps $4 --warmup 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape $1 $2 inflight $3: ms/step', round(d['ms_per_step'],3))"; }
for sh in 16 0; do
  for nf in 2 3 4 6; do run $sh grammar $nf 400; done
  for nf in 2 3 5; do run $sh mixed $nf 100; done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
