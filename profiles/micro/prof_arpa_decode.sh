# per-stage cycle counts of the token-list decoder on the ARPA workload (library rebuilt with -DRS_DECODE_PROFILE in a scratch copy)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02ae
mkdir -p /tmp/rsprof && cp -r rhasspy_speech_amd include /tmp/rsprof/
rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/decode_kernels.o
make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_DECODE_PROFILE > gpurun_out/r02ae/make.log 2>&1
cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_prof.so
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/librs_prof.so rhasspy_speech_amd/librhasspy_speech_hip.so
python bench.py --workload arpa --no-cpu-baseline --steps 2 --warmup 1 --inflight 1 2>&1 | grep "token-list" | tail -260 > gpurun_out/r02ae/arpa_stages.txt
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
