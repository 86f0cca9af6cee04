"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (csv output)."""
import csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-62s calls=%5s avg_us=%9.1f pct=%s" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
