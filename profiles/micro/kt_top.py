"""Prints the per-kernel averages of the newest rocprofv3 kernel_stats.csv under a directory (argv[1]); argv[2] = rows."""
import csv, sys
from pathlib import Path
f = max(Path(sys.argv[1]).rglob("*kernel_stats.csv"), key=lambda p: p.stat().st_mtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in list(csv.DictReader(open(f)))[:n]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f} us {r['Percentage']:>6s}%")
