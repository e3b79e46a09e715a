# Timing ablations of IvecAccumKernel (ivector_kernels.hip, RS_ACC_ABLATE / RS_ACC_TC): where the per-utterance statistics
# kernel spends its launch.  Scratch builds; the results of ablated runs are wrong by design.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/acc_ablate
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
mkdir -p /tmp/rsab && cp -r rhasspy_speech_amd include /tmp/rsab/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
date +%T > $OUT/progress.txt
timeout 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; echo "warm $? $(date +%T)" >> $OUT/progress.txt
for v in ${VARIANTS:-0,256 1,256 2,256 4,256 0,320}; do
  ab=${v%,*}; tc=${v#*,}
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/ivector_kernels.o
  timeout 300 make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA="-DRS_ACC_ABLATE=$ab -DRS_ACC_TC=$tc" > $OUT/make_${ab}_$tc.log 2>&1
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  rm -rf $OUT/kt_${ab}_$tc
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${ab}_$tc -- python bench.py --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/bench_${ab}_$tc.json 2> $OUT/bench_${ab}_$tc.log
  echo "$v done $(date +%T)" >> $OUT/progress.txt
  f=$(find $OUT/kt_${ab}_$tc -name "*kernel_stats.csv" | head -1)
  echo "ablate=$ab tc=$tc $(grep IvecAccumKernel $f | head -1 | awk -F'","|",|,' '{print "calls", $(NF-6), "avg_ns", $(NF-4)}')" >> $OUT/summary.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/summary.txt
