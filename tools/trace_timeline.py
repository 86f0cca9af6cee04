"""Prints the kernels of one timed bench step from a rocprofv3 kernel trace (csv): start relative to the step's first
kernel, duration, hardware queue and name - the picture of what runs beside what on the two streams."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
bwd = [i for i, r in enumerate(rows) if "composite_bwd" in r[2]]
first, last = bwd[-16 - 16], bwd[-16 - 9]          # the second-to-last timed step (16 launches of the breakdown steps come last)
i = first
while "preprocess_fwd_views" not in rows[i][2]:
    i -= 1
j = last
while "preprocess_bwd" not in rows[j][2]:
    j += 1
t0 = rows[i][0]
print("start ms  dur ms  queue  kernel")
for s, e, k, q in rows[i:j + 1]:
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", k)
    print("%8.3f %7.3f  q%s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, k.split("(")[0][:60]))
