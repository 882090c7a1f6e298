"""Per-launch averages of the counters of one rocprofv3 --pmc pass for the kernels whose name contains one of the filters (scripts/gpu_run.sh, step sq)."""
import collections, csv, glob, sys
tag, filt = sys.argv[1], sys.argv[2]
fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection*.csv") + glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection*.csv", recursive=True)
if not fs:
    print(tag, "no counter csv")
    sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(fs[0]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?").replace("void ", "").split("(")[0][:44]
        if not any(x in k for x in filt.split(",")):
            continue
        agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0))
        cnt[(k, row.get("Counter_Name"))] += 1
print("== pass", tag, "(per-launch averages)")
for k, v in agg.items():
    print(k)
    for n in sorted(v):
        print(f"   {n:28s} {v[n] / max(cnt[(k, n)], 1):14.5g}")
