#!/bin/bash
# the two LDA products (exact-FP32 GemmKernelDma, 40 columns) with 64- against 128-row tiles: RS_GEMM_NARROW_BM (-DRS_TUNING build)
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lda_bm}
mkdir -p $OUT
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -a rhasspy_speech_amd include /tmp/rstune/ && cp -a profiles/micro /tmp/rstune/profiles/
find /tmp/rstune/rhasspy_speech_amd/csrc -name '*.o' -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
cd /tmp && export TMPDIR=/tmp
for bm in 64 128; do
  RS_GEMM_NARROW_BM=$bm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$bm -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-figures --steps 10 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt$bm.log
  f=$(find $OUT/kt$bm -name "*kernel_stats.csv" | head -1)
  grep -E "GemmKernelDma" $f | awk -F, -v bm=$bm '{print "NARROW_BM=" bm, substr($1,1,44), "calls", $(NF-6), "avg_us", $(NF-4)/1000}'
  RS_GEMM_NARROW_BM=$bm python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-figures --steps 300 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],3))"
done
cp /tmp/librs_orig.so $GRAFT_REPO_ROOT/rhasspy_speech_amd/librhasspy_speech_hip.so
