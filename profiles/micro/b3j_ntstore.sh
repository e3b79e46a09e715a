# image stores of GemmKernelB3J as non-temporal stores (scratch build, -DRS_B3J_NT_STORE) against the default
cd "$GRAFT_REPO_ROOT"
run() { python bench.py --no-cpu-baseline --steps 150 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['stages_ms']['nnet'], d['roofline']['avg_launch_ms'])"; }
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
run default
mkdir -p /tmp/rsab && cp -r rhasspy_speech_amd include /tmp/rsab/
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/orig.so
rm -f /tmp/rsab/rhasspy_speech_amd/csrc/nnet_gemm_b3j.o
make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA=-DRS_B3J_NT_STORE > /tmp/mk.log 2>&1 || tail -3 /tmp/mk.log
cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
run nt_store
run nt_store
cp /tmp/orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
run default
