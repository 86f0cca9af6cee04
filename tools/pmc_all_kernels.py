"""Per-kernel vector-ALU time from the SQ counter passes of tools/pmc_composite.sh: for every kernel of the
profiled program, launches, vector instructions per launch, and SQ_ACTIVE_INST_VALU / 1024 SIMDs / 2.4 GHz per
launch - the time the kernel would take if nothing but the issue of its vector instructions limited it."""
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], d) not in seen and r["Counter_Name"] in ("SQ_WAVES", "SQ_WAIT_ANY"):
            seen.add((r["Dispatch_Id"], d)); calls[(k, d)] += 1
print("%-44s %5s %12s %10s %10s %10s" % ("kernel", "calls", "VALU inst", "valu us", "lds us", "busy us"))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_ACTIVE_INST_VALU", 0)):
    n = max(max(c for (kk, d), c in calls.items() if kk == k), 1)
    us = lambda name: v.get(name, 0.0) / n / 1024.0 / 2400.0
    print("%-44s %5d %12.4g %10.1f %10.1f %10.1f" % (k, n, v.get("SQ_INSTS_VALU", 0) / n, us("SQ_ACTIVE_INST_VALU"),
                                                     us("SQ_ACTIVE_INST_LDS"), v.get("SQ_BUSY_CYCLES", 0) / n / 2400.0 / 32))
