"""Per-launch GEMM durations of the last bench step from a rocprofv3 kernel trace (csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
g = [r for r in rows if "Gemm" in r["Kernel_Name"]][-12:]
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in g))
