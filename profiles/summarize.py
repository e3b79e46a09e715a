#!/usr/bin/env python3
"""Turns what profiles/collect.sh left under gpurun_out/<tag>_<workload>/ into the committed summaries profiles/<round>/<name>_*
(name = the workload, optionally with a version suffix, e.g. grammar or grammar_v2):
  *_kernel_stats.csv  rocprofv3 --kernel-trace --stats (copied as is)
  *_line.json         the bench.py line of the un-profiled run
  *_pmc.json          per kernel: mean counter value per launch for every --pmc pass (FETCH_SIZE / WRITE_SIZE in KB as
                      rocprofv3 reports them; bench.py applies the guide's correction)
usage: python profiles/summarize.py gpurun_out/r02_grammar profiles/r02 grammar
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

def strip_params(name: str) -> str:
    """Kernel name without its trailing parameter list (which may itself contain parentheses, as may the name)."""
    name = name.strip()
    if not name.endswith(")"):
        return name
    depth = 0
    for i in range(len(name) - 1, -1, -1):
        depth += name[i] == ")"
        depth -= name[i] == "("
        if depth == 0:
            return name[:i].strip()
    return name


src, dst, ver = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
dst.mkdir(parents=True, exist_ok=True)
def newest(pattern: str):
    """gpurun_out/ is merged across calls, so a directory may hold the files of earlier collections too (one set per profiled
    process id): the most recently written one is this collection's."""
    files = glob.glob(pattern, recursive=True)
    return max(files, key=lambda f: Path(f).stat().st_mtime) if files else None


stats = newest(str(src / "kt" / "**" / "*kernel_stats.csv"))
if stats:
    shutil.copy(stats, dst / f"{ver}_kernel_stats.csv")
line = (src / "bench_line.json").read_text().strip().splitlines()[-1]
json.loads(line)
(dst / f"{ver}_line.json").write_text(line + "\n")
for extra in ("all_pdfs", "one_call_in_flight"):          # the same bench with one switch changed (collect.sh)
    f = src / f"bench_line_{extra}.json"
    if f.exists() and f.read_text().strip():
        extra_line = f.read_text().strip().splitlines()[-1]
        json.loads(extra_line)
        (dst / f"{ver}_line_{extra}.json").write_text(extra_line + "\n")
kernels = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(str(src / "pmc_*"))):
    f = newest(str(Path(d) / "**" / "*counter_collection.csv"))
    if f is None:
        continue
    for r in csv.DictReader(open(f)):
        name = strip_params(r["Kernel_Name"])
        c = kernels[name][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
out = {"note": "rocprofv3 --pmc, separate passes (profiles/collect.sh), bench.py --steps 3 --warmup 1; FETCH_SIZE/WRITE_SIZE in KB "
               "as reported (FETCH_SIZE is doubled by bench.py per MI355X_MICROARCH.md: 128-B requests tallied at 64 B); SQ_* "
               "summed over the device",
       "kernels": {k: {c: {"launches": v[1], "mean": v[0] / v[1]} for c, v in cs.items()} for k, cs in kernels.items()}}
(dst / f"{ver}_pmc.json").write_text(json.dumps(out, indent=0))
print("wrote", sorted(p.name for p in dst.glob(f"{ver}_*")))
