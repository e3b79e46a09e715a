#!/bin/bash
# MfccKernel: cycles per phase of a few utterances (scratch build with -DRS_MFCC_PROFILE), headline batch, one call in flight
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_mfcc}
mkdir -p $OUT
rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -a rhasspy_speech_amd include /tmp/rsprof/
rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/feat_kernels.o
make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_MFCC_PROFILE > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
timeout 200 python bench.py --no-cpu-baseline --no-side-figures --steps 1 --warmup 1 --inflight 1 2>&1 | grep -o "mfcc row [^{]*" | tail -n 9 > $OUT/phases.txt
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/phases.txt
