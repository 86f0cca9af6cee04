"""Lists the idle intervals (no kernel running on any queue) longer than a threshold in a rocprofv3 kernel trace (csv)
of bench.py, with the kernels that end before and start after each: python tools/trace_idle.py DIR [min_us]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
short = lambda k: re.sub(r"\(anonymous namespace\)::|^void ", "", k).split("(")[0][:40]
bwd = [i for i, r in enumerate(rows) if "composite_bwd" in r[2]]
lo, hi = bwd[-16 - 40], bwd[-17]          # the last 40 timed composite_bwd launches (16 of the breakdown steps follow)
t0, busy, last, total = rows[lo][0], rows[lo][1], rows[lo][2], 0
for s, e, k in rows[lo + 1:hi + 1]:
    if s > busy:
        gap = (s - busy) / 1e3
        total += gap
        if gap >= min_us:
            print("%9.3f ms  idle %6.1f us  after %-40s before %s" % ((busy - t0) / 1e6, gap, short(last), short(k)))
    if e > busy:
        busy, last = e, k
print("window %.3f ms, idle %.3f ms in all" % ((busy - t0) / 1e6, total / 1e3))
