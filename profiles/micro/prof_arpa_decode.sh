# per-stage cycle counts of the large-graph search on the ARPA workload (library rebuilt with -DRS_DECODE_PROFILE in a scratch copy)
# usage (GPU box): bash profiles/micro/prof_arpa_decode.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-arpa_prof}
mkdir -p $OUT
rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -r rhasspy_speech_amd include /tmp/rsprof/
rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/decode_kernels.o
make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_DECODE_PROFILE > $OUT/make.log 2>&1
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
python bench.py --workload arpa --no-cpu-baseline --no-side-figures --steps 1 --warmup 1 --inflight 1 2>&1 | grep "block" | tail -256 > $OUT/arpa_stages.txt
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
sort -t: -k2 -n -r $OUT/arpa_stages.txt | sort -k4 -n -r | head -5
