#!/bin/bash
# PMC passes for the stall picture of one kernel family (separate passes; kernel-trace only, as gpurun requires).  usage: gpu_r05_pmc_sq.sh BATCH FILTER
mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-2048}; FILT=${2:-k_gwr_mol}
run() { tag=$1; shift
  rm -rf gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-roofline > gpurun_out/pmc_$tag.log 2>&1
  python - "$tag" "$FILT" <<'PY'
import csv, glob, collections, sys
tag, filt = sys.argv[1], sys.argv[2]
fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection*.csv")
if not fs:
    print(tag, "no counter csv"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
with open(fs[0]) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "?").replace("void ", "").split("(")[0][:44]
        if not any(x in k for x in filt.split(",")): continue
        agg[k][row.get("Counter_Name")] += float(row.get("Counter_Value", 0)); cnt[(k, row.get("Counter_Name"))] += 1
print("== pass", tag, "(per-launch averages)")
for k, v in agg.items():
    print(k)
    for n in sorted(v): print(f"   {n:28s} {v[n] / max(cnt[(k, n)], 1):14.5g}")
PY
  find gpurun_out/pmc_$tag -type f -size +2M -delete
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32
