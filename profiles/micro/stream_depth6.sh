# advances in flight of the streams workload beyond four: scratch builds with -DRS_STREAM_DEPTH=n -DRS_CONTEXT_SETS=n
cd "$GRAFT_REPO_ROOT"
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
for d in ${@:-4 6 8}; do
  rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -a rhasspy_speech_amd include /tmp/rsprof/
  find /tmp/rsprof/rhasspy_speech_amd/csrc -name "*.o" \( -name "stream.o" -o -name "engine.o" -o -name "api.o" -o -name "shard.o" \) -delete
  make -C /tmp/rsprof/rhasspy_speech_amd/csrc -j16 EXTRA="-DRS_STREAM_DEPTH=$d -DRS_CONTEXT_SETS=$d" > /tmp/rsprof/make.log 2>&1 || { tail /tmp/rsprof/make.log; continue; }
  cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  for rep in 1 2; do
    python bench.py --workload streams --steps 20 --warmup 3 --no-cpu-baseline --no-side-figures 2>/dev/null | tail -1 | python -c "
import sys, json
print('depth $d:', '%.2f ms/step' % json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
