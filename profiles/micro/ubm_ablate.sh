#!/bin/bash
# UbmPostMfmaKernel: scoring alone against scoring + selection (-DRS_TUNING build, RS_UBM_ABLATE=1)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ubm_ablate}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rstune && mkdir -p /tmp/rstune && cp -r rhasspy_speech_amd include profiles /tmp/rstune/
find /tmp/rstune/rhasspy_speech_amd/csrc -name "*.o" -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA="-DRS_TUNING" > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
cd /tmp && export TMPDIR=/tmp
for ab in 0 1; do
  RS_UBM_ABLATE=$ab timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$ab -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-figures --steps 10 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt$ab.log
  f=$(find $OUT/kt$ab -name "*kernel_stats.csv" | head -1)
  grep -E "UbmPost" $f | awk -F, -v ab=$ab '{print "RS_UBM_ABLATE=" ab, $1, "avg_us", $4/1000}' | cut -c1-120
done
cp /tmp/librs_orig.so $GRAFT_REPO_ROOT/rhasspy_speech_amd/librhasspy_speech_hip.so
