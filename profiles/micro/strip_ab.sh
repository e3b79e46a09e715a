#!/bin/bash
# GemmKernelB3J with / without the activation strip: bitwise tests, headline step, per-kernel time under rocprofv3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-strip_ab}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or offline_case or streaming_case" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
# (RS_GEMM_B3J_STRIP is a measurement switch: a -DRS_TUNING build of the library in a scratch copy, csrc/env.h)
rm -rf /tmp/rstune && mkdir -p /tmp/rstune/profiles && cp -r rhasspy_speech_amd include /tmp/rstune/ && cp -r profiles/micro /tmp/rstune/profiles/ && rm -f /tmp/rstune/rhasspy_speech_amd/csrc/*.o
make -C /tmp/rstune/rhasspy_speech_amd/csrc -j16 EXTRA=-DRS_TUNING > $OUT/make.log 2>&1 || { tail $OUT/make.log; exit 1; }
cp rhasspy_speech_amd/librhasspy_speech_hip.so /tmp/librs_orig.so; cp /tmp/rstune/rhasspy_speech_amd/librhasspy_speech_hip.so rhasspy_speech_amd/librhasspy_speech_hip.so
B="python bench.py --no-cpu-baseline --no-side-figures"
for strip in 1 0; do
  RS_GEMM_B3J_STRIP=$strip $B --steps 300 --warmup 20 > $OUT/line_strip$strip.json 2> $OUT/line_strip$strip.err
  python - <<PY
import json
d=json.loads(open("$OUT/line_strip$strip.json").read().strip().splitlines()[-1])
print("strip=$strip ms/step", round(d["ms_per_step"],3), "stages", {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "roofline", round(d["roofline"]["frac"],3), d["roofline"].get("avg_launch_ms"))
PY
  RS_GEMM_B3J_STRIP=$strip timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$strip -- $B --steps 20 --warmup 2 --inflight 1 > /dev/null 2> $OUT/kt$strip.log
  f=$(find $OUT/kt$strip -name "*kernel_stats.csv" | head -1)
  echo "strip=$strip kernel stats (one call in flight):"; python - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("GemmKernelB3", "RegDecode", "MfccKernel")):
        print("    %-70s calls %4s avg_us %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1000))
PY
done
cp /tmp/librs_orig.so rhasspy_speech_amd/librhasspy_speech_hip.so
