# token-list search: LDS prefix capacity (tokens per expansion chunk) against workgroups per CU; scratch builds with -DRS_PREFIX_CAP=n
cd "$GRAFT_REPO_ROOT"
run() { for k in 2 3; do python bench.py --workload arpa --no-cpu-baseline --steps 16 --inflight $k 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 inflight', $k, round(d['ms_per_step'],2), round(d['value']), round(d['stages_ms']['decode'],1))"; done; }
python bench.py --workload arpa --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
mkdir -p /tmp/rsab && cp -r rhasspy_speech_amd include /tmp/rsab/
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/orig.so
for cap in 8192 4096 2048; do
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/decode_kernels.o
  make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA=-DRS_PREFIX_CAP=$cap > /tmp/mk.log 2>&1 || tail -3 /tmp/mk.log
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  run cap=$cap
done
cp /tmp/orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
