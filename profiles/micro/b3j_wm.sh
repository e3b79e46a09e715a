# GemmKernelB3J shapes: launch time of the hidden layer and the headline step for RS_GEMM_B3J_WM = 1 | 2 (x scratch builds with extra flags)
# usage (GPU box): VARIANTS="-DX=1 -DX=2" WMS="1 2" bash profiles/micro/b3j_wm.sh
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/b3j_wm
mkdir -p $OUT; rm -f $OUT/summary.txt
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rsab && mkdir -p /tmp/rsab && cp -a rhasspy_speech_amd include /tmp/rsab/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for v in ${VARIANTS:--DRS_B3J_NOP=0}; do
  i=$((i+1))
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/nnet_gemm_b3j.o
  make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA="$v" > $OUT/make_$i.log 2>&1
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  for wm in ${WMS:-1 2}; do
    rm -rf $OUT/kt_${i}_$wm
    RS_GEMM_B3J_WM=$wm timeout -k 5 -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_${i}_$wm -- python bench.py --steps 12 --warmup 4 --inflight 1 --no-cpu-baseline --no-side-figures > /dev/null 2> $OUT/bench_${i}_$wm.log
    f=$(find $OUT/kt_${i}_$wm -name "*kernel_stats.csv" | head -1)
    step=$(RS_GEMM_B3J_WM=$wm timeout -k 5 -s KILL 300 python bench.py --no-cpu-baseline --no-side-figures --steps 300 --warmup 20 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
    echo "$v wm=$wm: $(grep GemmKernelB3J $f | awk -F'","|",|,' '{print "calls", $(NF-6), "avg_ns", $(NF-4)}' | tr '\n' ' ') headline step $step ms" >> $OUT/summary.txt
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/summary.txt
