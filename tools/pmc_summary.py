"""Sum the counters of a rocprofv3 --pmc run per kernel (csv output) and print the compositing kernels."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if any(s in k for s in sys.argv[2:] or ["composite"]):
        print(k[:70]); [print("   %-34s %.4g" % kv) for kv in sorted(v.items())]
