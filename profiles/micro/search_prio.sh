# the search's wave priority (s_setprio) against the pipelined step.  usage (GPU box): bash profiles/micro/search_prio.sh <out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-search_prio}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
for pr in 3 1 0; do
  rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -a rhasspy_speech_amd include /tmp/rsprof/
  rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/decode_reg.o
  make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_REG_PRIO=$pr > $OUT/make.log 2>&1
  cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  for infl in 1 4; do
    line=$(timeout 300 python bench.py --steps 200 --warmup 12 --inflight $infl --no-cpu-baseline --no-side-figures 2>/dev/null | tail -n 1)
    echo "s_setprio $pr, $infl in flight: $(echo "$line" | python -c 'import json,sys; l=json.loads(sys.stdin.read()); print("%.3f ms/step  stages %s" % (l["ms_per_step"], {k: round(v, 2) for k, v in l["stages_ms"].items()}))')" | tee -a $OUT/search_prio.txt
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
