# per-phase cycle counts of the live-state-table search (decode_live.hip) on the ARPA workload: library rebuilt with
# -DRS_DECODE_PROFILE in a scratch copy.  usage (GPU box): bash profiles/micro/prof_live_decode.sh <out dir under gpurun_out> [shape ...]
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-live_prof}; shift
SHAPES=${@:-1024}
mkdir -p $OUT
rm -rf /tmp/rsprof && mkdir -p /tmp/rsprof && cp -r rhasspy_speech_amd include /tmp/rsprof/
rm -f /tmp/rsprof/rhasspy_speech_amd/csrc/decode_live.o /tmp/rsprof/rhasspy_speech_amd/csrc/decode_kernels.o
make -C /tmp/rsprof/rhasspy_speech_amd/csrc -j16 EXTRA="-DRS_DECODE_PROFILE $RS_PROF_EXTRA" > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
cp /tmp/rsprof/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for sh in $SHAPES; do
  RS_LIVE_SHAPE=$sh python bench.py --workload arpa --no-cpu-baseline --no-side-figures --steps 1 --warmup 1 --inflight 1 2>&1 | grep "live block" | tail -256 > $OUT/stages_$sh.txt
  python - <<PY
import re
rows=[]
for l in open("$OUT/stages_$sh.txt"):
    m=re.match(r"live block (\d+): (\d+) cycles, (\d+) tokens, T=(\d+) phases cutoff (\d+) sweep (\d+) big (\d+) winners (\d+) closure (\d+) complete (\d+)", l)
    if m: rows.append([int(x) for x in m.groups()])
rows.sort(key=lambda r:-r[1])
names=["cutoff","sweep","big","winners","closure","complete"]
def show(r):
    tot=r[1]
    print("shape $sh block %d tokens %d total %.1f Mcyc:"%(r[0],r[2],tot/1e6)," ".join("%s %.1f%%"%(n,100*r[4+i]/tot) for i,n in enumerate(names)))
if rows:
    show(rows[0]); show(rows[len(rows)//2]); show(rows[-1])
    import numpy as np
    a=np.array(rows,dtype=np.float64)
    print("shape $sh sum over blocks: total %.0f Mcyc;"%(a[:,1].sum()/1e6)," ".join("%s %.1f%%"%(n,100*a[:,4+i].sum()/a[:,1].sum()) for i,n in enumerate(names)))
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
