"""profiles/traffic.json from the two PMC passes of tools/traffic_passes.sh.

Read bytes = 32 * RDREQ_32B + 128 * RDREQ_128B + 64 * (RDREQ - RDREQ_32B - RDREQ_128B): the L2's memory-side read
requests by size (exact; FETCH_SIZE tallies every request at 64 bytes and under-reports the 128-byte requests of
wide coalesced reads by half).  Written bytes = WRITE_SIZE (KiB) * 1024.  Per launch: the last launch of each kernel."""
import collections, csv, glob, json, re, sys
d = sys.argv[1]
out_path = sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json"


def last_launch(path):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        per.setdefault((r["Kernel_Name"], r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = {}
    for (k, _), v in per.items():
        last[k] = v
    return last


def short(k):
    m = re.search(r"([a-z_0-9]+)_kernel", k)
    return m.group(1) if m else k[:40]


rd = last_launch(glob.glob(d + "/rdreq/*/*counter_collection.csv")[0])
wr = last_launch(glob.glob(d + "/write/*/*counter_collection.csv")[0])
pairs = int(re.search(r"pairs (\d+)", open(d + "/rdreq.log").read()).group(1))
res = {}
for k, v in rd.items():
    n32, n128, n = v.get("TCC_EA0_RDREQ_32B_sum", 0), v.get("TCC_EA0_RDREQ_128B_sum", 0), v.get("TCC_EA0_RDREQ_sum", 0)
    res.setdefault(short(k), {})["read"] = int(32 * n32 + 128 * n128 + 64 * (n - n32 - n128))
for k, v in wr.items():
    res.setdefault(short(k), {})["write"] = int(v.get("WRITE_SIZE", 0) * 1024)
for k in res:
    res[k]["total"] = res[k].get("read", 0) + res[k].get("write", 0)
doc = {"_note": "HBM-side bytes of one launch of each kernel: 1 M Gaussians, 1920x1080, tile 16, SH degree 3, view 0 of "
                "bench.py's minibatch; rocprofv3 PMC, one pass per run (tools/traffic_passes.sh, tools/traffic_json.py: "
                "reads from the L2's memory-side read requests by size, writes from WRITE_SIZE).",
       "1000000_1920x1080_t16_deg3": {"pairs": pairs, "kernels": res}}
json.dump(doc, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if "composite" in k or "finalize" in k}, indent=1), "pairs", pairs)
