#!/bin/bash
# QHNet step: rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only -- MI355X_MICROARCH.md, HBM section)
MOL=${1:-16}
export TMPDIR=/tmp
CMD="python scripts/bench_qhnet.py --molecules $MOL --steps 2 --warmup 1"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_qh
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_qh -o qh -- python scripts/bench_qhnet.py --molecules $MOL --steps 10 --warmup 2 > gpurun_out/prof_qh.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $CMD > gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $CMD > gpurun_out/pmc_write.log 2>&1
python scripts/pmc_summary.py $MOL "$CMD" gpurun_out/pmc_traffic_qhnet.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_qh -type f -size +3M -delete
