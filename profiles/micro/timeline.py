"""What the device is doing in the steady state of the pipelined headline run: reads the newest rocprofv3 *kernel_trace.csv under
argv[1] and prints, for the middle half of the calls, how long each COMBINATION of kernel classes was on the device (sweep over the
dispatch start / end timestamps), per decode call (= number of MfccKernel dispatches in the window).

  gemm   = the layer GEMMs (device-filling)      mfcc / ubm = MfccKernel / UbmPostMfmaKernel (device-filling)
  search = RegDecodeKernel (one workgroup per utterance, latency-bound)      small = everything else (iVector chain, copies, ...)
"""
import csv
import os
import sys
from collections import defaultdict
from pathlib import Path


STREAMS = os.environ.get("TIMELINE_STREAMS") == "1"


def klass(name):
    if "Ivec" in name:            # (only apart from "small" where it matters: the streams workload)
        return "ivec" if STREAMS else "small"
    if "Gemm" in name:
        return "gemm"
    if "MfccKernel" in name:
        return "mfcc"
    if "UbmPost" in name:
        return "ubm"
    if "DecodeKernel" in name or "DecodeExact" in name:
        return "search"
    return "small"


def main():
    f = max(Path(sys.argv[1]).rglob("*kernel_trace.csv"), key=lambda p: p.stat().st_mtime)
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
    # the window: from the start of the call a quarter into the run to the start of the one three quarters in (model set-up, warm-up
    # and the drain at the end stay outside)
    mf = [s for s, e, n in ev if "MfccKernel" in n]
    lo, hi = mf[len(mf) // 4], mf[3 * len(mf) // 4]
    calls = 3 * len(mf) // 4 - len(mf) // 4
    points = []
    for s, e, n in ev:
        s2, e2 = max(s, lo), min(e, hi)
        if s2 >= e2:
            continue
        k = klass(n)
        points.append((s2, 1, k))
        points.append((e2, -1, k))
    points.sort(key=lambda p: (p[0], p[1]))
    active = defaultdict(int)
    last = lo
    dur = defaultdict(int)
    for t, d, k in points:
        if t > last:
            key = "+".join(sorted(c for c, v in active.items() if v > 0)) or "idle"
            dur[key] += t - last
            last = t
        active[k] += d
    if hi > last:
        dur["idle"] += hi - last
    span = hi - lo
    print(f"{f.name}: window {span / 1e6:.2f} ms, {calls} decode calls -> {span / 1e3 / max(calls, 1):.1f} us per call")
    for key, v in sorted(dur.items(), key=lambda kv: -kv[1]):
        print(f"  {key:32s} {v / 1e3 / max(calls, 1):9.1f} us per call  {100.0 * v / span:5.1f} %")
    filling = sum(v for k, v in dur.items() if any(c in k.split("+") for c in ("gemm", "mfcc", "ubm")))
    print(f"  a device-filling kernel is running {100.0 * filling / span:.1f} % of the time")
    # kernel time per call by class, for comparison with the window
    per = defaultdict(int)
    for s, e, n in ev:
        if lo <= s < hi:
            per[klass(n)] += e - s
    print("  kernel time per call by class (us): " + ", ".join(f"{k} {v / 1e3 / max(calls, 1):.1f}" for k, v in sorted(per.items())))


if __name__ == "__main__":
    main()
