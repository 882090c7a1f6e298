#!/bin/bash
# L2 request counters of the message kernels on the round-4 build (one PMC pass, kernel trace only): TCC_HIT_sum + TCC_MISS_sum = requests that reached the L2,
# i.e. the gather / store traffic between the CUs and the L2 that DESIGN 7 prices (x 128 B per request as an upper bound, x 64 B as a lower one).
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tcc
timeout -k 5 300 rocprofv3 --pmc ${PMC_SET:-TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum} --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_tcc.log 2>&1
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_tcc/*counter_collection*.csv")
if not fs:
    print("no counter csv"); print(open("gpurun_out/pmc_tcc.log").read()[-1500:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(fs[0]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?").replace("void ", "").split("(")[0][:40]
        agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0)); cnt[(k, row.get("Counter_Name"))] += 1
names = sorted({c for v in agg.values() for c in v})
out = ["# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline  (B = 2048, round-4 build)",
       "# per-launch averages of the L2 (TCC) request counters, the 12 kernels with the most requests", f"{'kernel':40s} " + " ".join(f"{n:>16s}" for n in names) + "   launches"]
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    out.append(f"{k:40s} " + " ".join(f"{v[n] / max(cnt[(k, n)], 1):16.4g}" for n in names) + f"   {max(cnt[(k, n)] for n in names)}")
open("gpurun_out/r04_pmc_" + __import__("os").environ.get("PMC_TAG", "tcc") + ".txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find gpurun_out/pmc_tcc -type f -size +2M -delete
