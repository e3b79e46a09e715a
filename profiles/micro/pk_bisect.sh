# Does the interference of DESIGN.md section 5 (wrong features when the kernels are built with the compiler's vectorizers on and run
# beside another decode call) still reproduce, and which transformation of which file does it take?  Scratch builds of the library
# with the flags of csrc/Makefile's NOPACK changed (all four files it applies to / feat_kernels.hip only; SLP vectorizer / loop
# vectorizer separately), the concurrent-call stress test on each.
#   usage (GPU box): bash profiles/micro/pk_bisect.sh <out dir under gpurun_out> [iterations]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pk_bisect}; IT=${2:-40}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
run() {   # label, feat-only (0/1), NOPACK value
  local label=$1 featonly=$2 flags=$3
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o ivector_kernels.o nnet_kernels.o decode_kernels.o
    if [ $featonly = 1 ]; then make NOPACK="$flags" feat_kernels.o && make; else make NOPACK="$flags"; fi ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "$label: $(timeout 600 python profiles/micro/stress_same_model.py $IT 4 2>&1 | tail -3 | cut -c1-260 | tr '\n' ' ')"
}
run_others() {
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o ivector_kernels.o nnet_kernels.o decode_kernels.o
    make feat_kernels.o && make NOPACK= ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "the other three files with both vectorizers, feat_kernels.hip as shipped: $(timeout 600 python profiles/micro/stress_same_model.py $IT 4 2>&1 | tail -3 | cut -c1-260 | tr '\n' ' ')"
}
if [ "${3:-}" != poison ]; then
{
run "as shipped (-fno-slp-vectorize -fno-vectorize)" 0 "-fno-slp-vectorize -fno-vectorize"
run "feat_kernels.hip alone with both vectorizers" 1 ""
run "feat_kernels.hip alone with the loop vectorizer only (-fno-slp-vectorize)" 1 "-fno-slp-vectorize"
run "feat_kernels.hip alone with the SLP vectorizer only (-fno-vectorize)" 1 "-fno-vectorize"
run_others
} > $OUT/result.txt 2>&1
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/result.txt
fi
# ... and the same scratch build of feat_kernels.hip (SLP vectorizer on) with ONE call in flight but NaN patterns left in LDS and in the
# register files in front of every kernel (RS_LDS_POISON=1, profiles/micro/lds_poison_check.py): a kernel that reads a register or LDS
# word it never wrote changes its result with nothing else running
if [ "${3:-}" = poison ]; then
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o && make NOPACK="-fno-vectorize" feat_kernels.o && make ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "feat_kernels.hip with the SLP vectorizer, one call in flight, poisoned LDS / registers:" >> $OUT/result.txt
  timeout 600 python profiles/micro/lds_poison_check.py 2>&1 | tail -16 >> $OUT/result.txt
  cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
  tail -17 $OUT/result.txt
fi
