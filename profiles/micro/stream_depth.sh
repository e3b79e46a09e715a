# advances in flight (arena sets) of the streams workload: scratch builds with -DRS_STREAM_DEPTH=n
cd "$GRAFT_REPO_ROOT"
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
python bench.py --workload streams --steps 2 --warmup 1 --no-cpu-baseline --no-side-figures > /dev/null 2>&1
for d in 3 4 2; do
  rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -a rhasspy_speech_amd include /tmp/rsprof/
  rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/stream.o
  make -C /tmp/rsprof/rhasspy_speech_amd/csrc EXTRA=-DRS_STREAM_DEPTH=$d > /dev/null 2>&1
  cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  for rep in 1 2; do
    RS_STREAMS_TRACE=1 python bench.py --workload streams --steps 20 --warmup 3 --no-cpu-baseline --no-side-figures 2>&1 | grep -E "^streams|ms_per_step" | tail -2 | python -c "
import sys, json
ls = sys.stdin.read().strip().splitlines()
print('depth $d:', '%.2f ms/step' % json.loads(ls[-1])['ms_per_step'], '|', ls[0][:110])"
  done
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
