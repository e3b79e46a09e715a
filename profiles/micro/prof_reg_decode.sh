# per-phase cycle counts of the grammar-graph search (RegDecodeKernel) on the headline workload: library rebuilt with
# -DRS_DECODE_PROFILE in a scratch copy; utterance 0 prints its cycles per frame.  usage (GPU box): bash profiles/micro/prof_reg_decode.sh <out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-reg_prof}
mkdir -p $OUT
rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -a rhasspy_speech_amd include /tmp/rsprof/
rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/decode_reg.o
make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_DECODE_PROFILE > $OUT/make.log 2>&1
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for nt in auto 512; do
  for infl in 1 4; do
    if [ $nt = auto ]; then unset RS_REG_NT; else export RS_REG_NT=$nt; fi
    python bench.py --no-cpu-baseline --no-side-figures --steps 2 --warmup 1 --inflight $infl 2>&1 | grep "reg decode" | tail -n 8 > $OUT/reg_phases_${nt}_inflight$infl.txt
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
tail -n 3 $OUT/reg_phases_*.txt
