# Per-workgroup timeline of one hidden-layer launch of GemmKernelB3J (library rebuilt with -DRS_B3J_TRACE in a scratch copy):
# start / end of k loop / end of epilogue (s_memrealtime, 10 ns ticks) and the CU each workgroup ran on.
# usage (GPU box): bash profiles/micro/b3j_trace.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-b3j_trace}
mkdir -p $OUT
rm -rf /tmp/rstr && mkdir -p /tmp/rstr && cp -a rhasspy_speech_amd include /tmp/rstr/
rm -f /tmp/rstr/rhasspy_speech_amd/csrc/nnet_gemm_b3j.o
make -C /tmp/rstr/rhasspy_speech_amd/csrc EXTRA="-DRS_B3J_TRACE -DRS_B3J_ABLATE=${ABLATE:-0}" > $OUT/make.log 2>&1
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rstr/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
RS_B3J_TRACE_FILE=$OUT/trace.txt python bench.py --no-cpu-baseline --no-side-figures --steps 8 --warmup 4 --inflight 1 > $OUT/bench.json 2> $OUT/bench.log
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
head -3 $OUT/trace.txt; wc -l $OUT/trace.txt
