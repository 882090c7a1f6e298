#!/bin/bash
# (1) distributed plumbing of bench.py with 2 ranks on the single GPU (gloo; RCCL rejects duplicate devices)
# (2) PMC counters for HBM traffic of the dominant kernels (separate passes, as the microarch guide prescribes)
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== 2-rank bench over gloo on one GPU"
NQ_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 256 --no-roofline 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/bench_2rank_gloo.log
echo "== B=32 (reference batch size) and B=256"
for b in 32 256; do timeout 300 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --no-roofline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " batch=$b"; done | tee gpurun_out/bench_small_batches.log
echo "== PMC pass 1: FETCH_SIZE"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --batch 1024 --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1
echo "== PMC pass 2: WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --batch 1024 --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
ls gpurun_out/pmc_fetch gpurun_out/pmc_write
python - <<'PY'
import csv, glob, collections
for tag in ("fetch", "write"):
    fs = glob.glob(f"gpurun_out/pmc_{tag}/*counter_collection*.csv")
    if not fs:
        print(tag, "no counter csv", glob.glob(f"gpurun_out/pmc_{tag}/*")); continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    with open(fs[0]) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")[:60]
            agg[(k, row.get("Counter_Name"))][0] += float(row.get("Counter_Value", 0)); agg[(k, row.get("Counter_Name"))][1] += 1
    print("==", tag, fs[0])
    for (k, c), (v, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:14]:
        print(f"{k:60s} {c:12s} total {v:14.1f} launches {n:5d} per-launch {v/n:12.1f}")
PY
find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f -size +3M -delete
