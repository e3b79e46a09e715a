# Launch time of GemmKernelB3J for scratch builds with extra compile flags (one per variant), e.g.
#   VARIANTS="-DRS_B3J_PRIO=0 -DRS_B3J_PRIO=1" bash profiles/micro/b3j_variant.sh      (results stay correct unless a flag says otherwise)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/b3j_variant
mkdir -p $OUT; rm -f $OUT/summary.txt
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rsab && mkdir -p /tmp/rsab && cp -a rhasspy_speech_amd include /tmp/rsab/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for v in $VARIANTS; do
  i=$((i+1))
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/nnet_gemm_b3j.o
  make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA="$v" > $OUT/make_$i.log 2>&1
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  rm -rf $OUT/kt_$i
  timeout -k 5 -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$i -- python bench.py --steps 12 --warmup 4 --inflight 1 --no-cpu-baseline --no-side-figures > /dev/null 2> $OUT/bench_$i.log
  f=$(find $OUT/kt_$i -name "*kernel_stats.csv" | head -1)
  step=$(timeout -k 5 -s KILL 300 python bench.py --no-cpu-baseline --no-side-figures 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
  echo "$v: $(grep GemmKernelB3J $f | head -1 | awk -F'","|",|,' '{print "calls", $(NF-6), "avg_ns", $(NF-4)}') headline step $step ms" >> $OUT/summary.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/summary.txt
