# Timing ablations of GemmKernelB3J (nnet_gemm_b3j.hip, RS_B3J_ABLATE): which of the weight stream, the activation stream, the
# matrix cores and the stage barrier the launch time follows.  Scratch builds; the results of ablated runs are wrong by design.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/b3j_ablate
mkdir -p $OUT; rm -f $OUT/summary.txt
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
mkdir -p /tmp/rsab && cp -a rhasspy_speech_amd include /tmp/rsab/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for v in ${VARIANTS:-0 64}; do
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/nnet_gemm_b3j.o
  make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA=-DRS_B3J_ABLATE=$v > $OUT/make_$v.log 2>&1
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  rm -rf $OUT/kt_$v
  RS_GEMM_B3J_ONE_PER_CU=${ONE:-0} timeout -k 5 -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$v -- python bench.py --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --no-side-figures > $OUT/bench_$v.json 2> $OUT/bench_$v.log
  f=$(find $OUT/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "ablate=$v $(grep GemmKernelB3J $f | head -1 | awk -F'","|",|,' '{print "calls", $(NF-6), "avg_ns", $(NF-4)}')" >> $OUT/summary.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/summary.txt
