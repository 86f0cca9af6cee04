"""Reads a rocprofv3 kernel trace (csv) and reports, for the last `steps` bench steps, how much of the wall
time the GPU ran the compositing kernels, other kernels only, or nothing."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# window: from the start of the 17th-from-last composite_bwd to the end (2 breakdown steps = 16 launches are last)
bw = [r for r in rows if "composite_bwd" in r[2]]
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lo = bw[-16 - n_timed][0]; hi = bw[-17][1]
ev = []
for s, e, k in rows:
    if e <= lo or s >= hi: continue
    c = 1 if "composite" in k else 0
    ev.append((max(s, lo), 1, c)); ev.append((min(e, hi), -1, c))
ev.sort()
t = lo; act = [0, 0]; acc = {"composite": 0, "other only": 0, "idle": 0}
for ts, d, c in ev:
    dt = ts - t
    if act[1] > 0: acc["composite"] += dt
    elif act[0] > 0: acc["other only"] += dt
    else: acc["idle"] += dt
    act[c] += d; t = ts
tot = hi - lo
print("window %.3f ms over %d composite_bwd launches" % (tot / 1e6, n_timed))
for k, v in acc.items(): print("  %-12s %8.3f ms  %5.1f %%" % (k, v / 1e6, 100.0 * v / tot))
