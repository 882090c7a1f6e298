"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes)
into per-kernel KB per launch.  rocprofv3 reports both in KB; on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x (guide, section HBM) -- the doubling is applied by the consumer (bench.py), the raw values are kept here."""
import collections
import csv
import glob
import json
import sys

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cmd = sys.argv[2] if len(sys.argv) > 2 else "python bench.py --batch %d" % batch
dest = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/pmc_traffic.json"
out = {"batch": batch, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) around `%s`" % cmd,
       "units": "KB per launch (raw counter values)", "kernels": {}}
for tag, key in (("fetch", "fetch_kb_per_launch"), ("write", "write_kb_per_launch")):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection*.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(fs[0]) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")
            k = k.replace("void ", "").split("(")[0]
            agg[k][0] += float(row.get("Counter_Value", 0))
            agg[k][1] += 1
    for k, (v, n) in agg.items():
        out["kernels"].setdefault(k, {})[key] = v / n
        out["kernels"][k]["launches_" + tag] = n
top = sorted(out["kernels"].items(), key=lambda kv: -(kv[1].get("fetch_kb_per_launch", 0) * kv[1].get("launches_fetch", 0)
                                                      + kv[1].get("write_kb_per_launch", 0) * kv[1].get("launches_write", 0)))[:25]
out["kernels"] = dict(top)
json.dump(out, open(dest, "w"), indent=1)
for k, v in top[:12]:
    print(f"{k[:70]:70s} fetch {v.get('fetch_kb_per_launch', 0) / 1e6:8.3f} GB(raw)  write {v.get('write_kb_per_launch', 0) / 1e6:8.3f} GB per launch")
