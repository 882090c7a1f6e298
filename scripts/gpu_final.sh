#!/bin/bash
# Full measured round: tests, smoke, bench (+cpu baseline), rocprof stats, PMC traffic passes
BATCH=${1:-1024}; TAG=${2:-final}
bash scripts/gpu_round.sh $BATCH $TAG
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline --no-roofline > gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline --no-roofline > gpurun_out/pmc_write.log 2>&1
python scripts/pmc_summary.py $BATCH
find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f -size +3M -delete
