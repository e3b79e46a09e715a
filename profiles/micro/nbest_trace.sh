#!/bin/bash
# where an n-best call's time goes, alone and with four calls in flight: a -DRS_TUNING build (RS_LATTICE_TRACE=1 prints the tail's
# sections per call).  usage (GPU box): bash profiles/micro/nbest_trace.sh <out dir under gpurun_out>
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-nbest_trace}
mkdir -p $OUT
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so
rm -rf /tmp/rstune && mkdir -p /tmp/rstune && cp -r rhasspy_speech_amd include profiles /tmp/rstune/
find /tmp/rstune/rhasspy_speech_amd/csrc -name "*.o" -delete
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j32 EXTRA="-DRS_TUNING" > $OUT/make.log 2>&1 || { tail -20 $OUT/make.log; exit 1; }
cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
for k in ${KS:-1 4}; do
  RS_LATTICE_TRACE=1 python profiles/micro/nbest_inflight.py $k > $OUT/trace_k$k.txt 2>&1
  python - <<PY
import re, numpy as np
rows = []
for l in open("$OUT/trace_k$k.txt"):
    m = re.search(r"kernel \+ count ([\d.]+) ms.*to the host ([\d.]+) ms, grouping ([\d.]+) ms, (\d+) utterances on (\d+) threads ([\d.]+) ms", l)
    if m: rows.append([float(m.group(i)) for i in (1, 2, 3, 6)])
    elif "in flight" in l: print(l.strip())
a = np.array(rows[len(rows) // 3:])
print("  $k in flight, tail sections (mean of %d calls): kernel + count %.2f, arcs to the host %.2f, grouping %.2f, per-utterance jobs %.2f ms" % (len(a), *a.mean(0)))
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
