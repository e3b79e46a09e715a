#!/bin/bash
# MfccKernel shapes: 4 waves per workgroup with the FFT plan read from memory (0, default) against 4 / 8 / 16 waves with the plan in LDS
# (RS_MFCC_SHAPE, -DRS_TUNING build): kernel time alone (one call in flight) and the pipelined step
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-mfcc_shape}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
cd /tmp && export TMPDIR=/tmp
for sh in ${SHAPES:-0 4 8 16}; do
  RS_MFCC_SHAPE=$sh timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$sh -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-figures --steps 10 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt$sh.log
  f=$(find $OUT/kt$sh -name "*kernel_stats.csv" | head -1)
  grep -E "MfccKernel" $f | awk -F, -v sh=$sh '{print "SHAPE=" sh, substr($1,1,40), "avg_us", $(NF-4)/1000}'
  RS_MFCC_SHAPE=$sh python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-figures --steps 400 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step', round(d['ms_per_step'],3))"
done
cp /tmp/librs_orig.so $GRAFT_REPO_ROOT/rhasspy_speech_amd/librhasspy_speech_hip.so
