"""Reads the per-workgroup records profiles/micro/b3j_trace.sh leaves: the launch's timeline per CU."""
import sys, collections
import numpy as np
rows = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
hdr = open(sys.argv[1]).readline().strip()
b = np.array([[int(r[0]), int(r[1]), int(r[2]), int(r[3])] for r in rows], dtype=np.int64)
hw = [int(r[4], 16) for r in rows]
t0 = b[:, 1].min()
start, loop_end, end = (b[:, 1] - t0) / 100.0, (b[:, 2] - t0) / 100.0, (b[:, 3] - t0) / 100.0     # us
print(hdr)
print(f"launch span {end.max():.1f} us; workgroups {len(b)}")
nfirst = int(hdr.split("nfirst")[1].split()[0]); nbig = int(hdr.split("nbig")[1].split()[0])
kind = np.array(["small" if (i < nfirst or i >= nfirst + nbig) else "big" for i in b[:, 0]])
for k in ("small", "big"):
    m = kind == k
    if m.any():
        print(f"{k:5s}: n {m.sum():4d}  start {start[m].mean():6.1f} (min {start[m].min():.1f} max {start[m].max():.1f})  "
              f"k loop {np.mean(loop_end[m] - start[m]):6.1f} us (min {np.min(loop_end[m] - start[m]):.1f} max {np.max(loop_end[m] - start[m]):.1f})  "
              f"epilogue {np.mean(end[m] - loop_end[m]):5.1f} us (min {np.min(end[m] - loop_end[m]):.1f} max {np.max(end[m] - loop_end[m]):.1f})")
# per CU: (xcc, se, cu) from HW_ID: cu_id bits 11:8, sh 12, se 15:13 (gfx9 layout)
cu = collections.defaultdict(list)
for i, h in enumerate(hw):
    xcc = h >> 32; w = h & 0xFFFFFFFF
    cu[(xcc & 0xF, (w >> 13) & 7, (w >> 12) & 1, (w >> 8) & 0xF)].append(i)
if len(rows[0]) > 5:
    clk = np.array([int(r[5]) for r in rows], dtype=np.float64)
    ghz = clk / ((b[:, 2] - b[:, 1]) * 10.0)            # shader clocks per ns of the k loop
    print(f"shader clock during the k loops: mean {ghz.mean():.3f} GHz (min {ghz.min():.3f} max {ghz.max():.3f})")
print(f"distinct CUs {len(cu)}; workgroups per CU: {collections.Counter(len(v) for v in cu.values())}")
busy = []
for key, idx in cu.items():
    iv = sorted((start[i], end[i]) for i in idx)
    busy.append(sum(e - s for s, e in iv))
print(f"sum of workgroup lifetimes per CU: mean {np.mean(busy):.1f} us (two slots x {end.max():.1f} = {2 * end.max():.1f})")
# how many workgroups are in their epilogue at a time
ts = np.linspace(0, end.max(), 200)
ne = [(int(((loop_end <= t) & (end > t)).sum()), int(((start <= t) & (loop_end > t)).sum())) for t in ts]
print("time us: workgroups in k loop / in epilogue")
for t, (e, l) in list(zip(ts, ne))[::10]:
    print(f"  {t:6.1f}: {l:4d} / {e:4d}")
for key in list(cu)[:3]:
    print("CU", key, [(kind[i], round(start[i], 1), round(loop_end[i], 1), round(end[i], 1)) for i in sorted(cu[key], key=lambda i: start[i])])
