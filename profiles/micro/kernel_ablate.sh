# Timing ablations through scratch builds: FILE=<csrc file without suffix> KERNEL=<name to grep in the kernel stats>
# VARIANTS="<-D flags of build 1>;<-D flags of build 2>;..." bash profiles/micro/kernel_ablate.sh
# (the results of ablated runs are wrong by design; the shipped library is restored at the end)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ablate_$KERNEL
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
mkdir -p /tmp/rsab && cp -r rhasspy_speech_amd include /tmp/rsab/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 240 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
i=0
IFS=';' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  i=$((i+1))
  rm -f /tmp/rsab/rhasspy_speech_amd/csrc/$FILE.o
  timeout 300 make -C /tmp/rsab/rhasspy_speech_amd/csrc EXTRA="$v" > $OUT/make_$i.log 2>&1
  cp /tmp/rsab/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
  rm -rf $OUT/kt_$i
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$i -- python bench.py --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/bench_$i.log
  echo "[$v] $(python profiles/micro/kt_top.py $OUT/kt_$i 40 | grep "$KERNEL" | head -1)" >> $OUT/summary.txt
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
cat $OUT/summary.txt
