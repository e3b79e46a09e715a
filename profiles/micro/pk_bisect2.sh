# Which of the SLP vectorizer's packed operations in MfccKernel does the interference need?  Scratch builds of feat_kernels.hip with the
# SLP vectorizer on and its profitability threshold raised step by step (fewer and fewer packed operations: 30 / 26 / 22 / 11 / 0 at
# -slp-threshold 0 / 2 / 4 / 6 / 12), the concurrent-call stress test on each.
#   usage (GPU box): bash profiles/micro/pk_bisect2.sh <out dir under gpurun_out> [iterations] [thresholds...]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pk_bisect2}; IT=${2:-30}
shift; shift
TH=${@:-0 4 6 12}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for t in $TH; do
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o && make NOPACK="-fno-vectorize -mllvm -slp-threshold=$t $EXTRA_FEAT" feat_kernels.o && make ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "slp-threshold $t $EXTRA_FEAT: $(timeout 600 python profiles/micro/stress_same_model.py $IT 4 2>&1 | tail -2 | cut -c1-200 | tr '\n' ' ')" | tee -a $OUT/result.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
