"""Prints the kernels of one step of `bench.py --views 1` from a rocprofv3 kernel trace (csv): start relative to the
step's first kernel, duration, the idle time before it (no kernel running on any queue) and the hardware queue."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # which step, counted from the end of the run
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
ends = [i for i, r in enumerate(rows) if "preprocess_bwd" in r[2]]
lo, hi = ends[-back - 1] + 1, ends[-back]
t0, busy_until, idle = rows[lo][0], rows[lo][0], 0
print("start ms  dur ms  idle before  queue  kernel")
for s, e, k, q in rows[lo:hi + 1]:
    gap = max(0, s - busy_until)
    idle += gap
    busy_until = max(busy_until, e)
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
    print("%8.3f %7.3f %9.3f    q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, q, k.split("(")[0][:60]))
print("step %.3f ms, of which idle %.3f ms" % ((rows[hi][1] - t0) / 1e6, idle / 1e6))
