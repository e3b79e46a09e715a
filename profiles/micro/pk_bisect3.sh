# Which REGION of MfccKernel must carry SLP-formed packed operations for the interference?  Scratch builds (SLP vectorizer on) with one
# region moved into a noinline + optnone helper (profiles/micro/pk_bisect3.py): k1 = the length-4 transform of the FFT, pw = the
# power-spectrum loop; the concurrent-call stress test on each.   usage (GPU box): bash profiles/micro/pk_bisect3.sh <out> [iterations] [variants...]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-pk_bisect3}; IT=${2:-30}
shift; shift
VARS=${@:-none k1 pw}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for v in $VARS; do
  rm -rf /tmp/rspk && mkdir -p /tmp/rspk && cp -r rhasspy_speech_amd include /tmp/rspk/
  python profiles/micro/pk_bisect3.py /tmp/rspk/rhasspy_speech_amd/csrc/feat_kernels.hip $v
  ( cd /tmp/rspk/rhasspy_speech_amd/csrc && rm -f feat_kernels.o && make NOPACK="-fno-vectorize $EXTRA_FEAT" feat_kernels.o && make ) > /tmp/rspk/make.log 2>&1
  cp /tmp/rspk/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  echo "SLP on, variant $v $EXTRA_FEAT: $(timeout 600 python profiles/micro/stress_same_model.py $IT 4 2>&1 | tail -2 | cut -c1-60,170-260 | tr '\n' ' ')" | tee -a $OUT/result.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
