#!/bin/bash
# PMC passes for the stall picture of the fused message kernels (separate passes; kernel-trace only, as gpurun requires)
mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-1024}
run() { tag=$1; shift
  rm -rf gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-roofline > gpurun_out/pmc_$tag.log 2>&1
  python - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection*.csv")
if not fs:
    print(tag, "no counter csv"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(fs[0]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?").replace("void ", "").split("(")[0][:44]
        agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0)); cnt[(k, row.get("Counter_Name"))] += 1
names = sorted({c for v in agg.values() for c in v})
print("== pass", tag, "(per-launch averages)")
print(f"{'kernel':44s} " + " ".join(f"{n[-18:]:>18s}" for n in names))
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:12]
for k, v in rows:
    print(f"{k:44s} " + " ".join(f"{v[n] / max(cnt[(k, n)], 1):18.4g}" for n in names))
PY
  find gpurun_out/pmc_$tag -type f -size +2M -delete
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
