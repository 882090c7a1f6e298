#!/bin/bash
# bench.py (HIP-event roofline) and the rocprofv3 kernel stats of the same command, back to back
export TMPDIR=/tmp
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_pair.log 2>/dev/null
rm -rf gpurun_out/prof_pair
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pair -o painn -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rocprof_pair.log 2>&1
find gpurun_out/prof_pair -type f ! -name "*stats*" -size +2M -delete
tail -1 gpurun_out/bench_pair.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench', round(d['value']), round(d['ms_per_step'],2), 'dual avg ms', round(r['avg_launch_ms'],4))
"
f=$(find gpurun_out/prof_pair -name "*kernel_stats*.csv" | head -1); grep "k_msgf_rev<true" $f | cut -c1-140
tail -1 gpurun_out/rocprof_pair.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('under rocprof: dual avg ms', round(d['roofline']['avg_launch_ms'],4))
except Exception as e: print('n/a', e)
"
