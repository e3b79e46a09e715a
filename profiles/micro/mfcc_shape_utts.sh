#!/bin/bash
# MfccKernel shape (RS_MFCC_SHAPE 0 / 16) against the batch size of the headline workload, four calls in flight (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfcc_shape_utts}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
run() { RS_MFCC_SHAPE=$1 python bench.py --utts $2 --no-cpu-baseline --no-side-figures --steps $3 --warmup 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape $1 utts $2: ms/step', round(d['ms_per_step'],3), 'per 256 utts', round(d['ms_per_step']*256/$2,3))"; }
for u in 64 128 192 256 384 512; do for sh in 16 0; do run $sh $u 300; done; done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
