#!/bin/bash
# the 16-frame LDS-plan MfccKernel shape against RS_MFCC_EXCLUSIVE=1 (16 frames per workgroup + all of a CU's LDS requested: no layer-GEMM
# workgroup can share the CU) in the pipelined headline step (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfcc_excl}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for e in 0 1 0 1; do
  RS_MFCC_EXCLUSIVE=$e python bench.py --no-cpu-baseline --no-side-figures --steps 400 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exclusive $e: ms/step', round(d['ms_per_step'],3))"
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
