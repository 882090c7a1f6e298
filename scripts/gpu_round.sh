#!/bin/bash
# One GPU-box visit: tests, smoke, short bench, rocprof kernel stats. Logs land in gpurun_out/.
# usage: scripts/gpu_round.sh [batch] [tag]
BATCH=${1:-256}
TAG=${2:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; (rocminfo | grep -E "Marketing Name|gfx9" | head -4) 2>&1
echo "== unit tests (gemm / graph / loss / adamw)"
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --tb=short -k "library or gemm or weight_grad or graph or loss_kernel" 2>&1 | tail -40 | tee gpurun_out/pytest_unit_$TAG.log
echo "== engine tests"
timeout 1200 python -m pytest tests/test_engine_gpu.py -m gpu -q --tb=short -k "golden or fused or properties or training" 2>&1 | tail -60 | tee gpurun_out/pytest_engine_$TAG.log
echo "== smoke"
timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/smoke_$TAG.log
echo "== bench"
timeout 900 python bench.py --steps 5 --warmup 2 --batch $BATCH $BENCH_EXTRA 2>&1 | tail -3 | tee gpurun_out/bench_$TAG.log
echo "== rocprof"
rm -rf gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o painn -- python bench.py --steps 3 --warmup 1 --batch $BATCH --no-cpu-baseline --no-roofline > gpurun_out/rocprof_$TAG.log 2>&1
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -25 "$f"
# keep only the small summaries
find gpurun_out/prof_$TAG -type f ! -name "*stats*" -size +2M -delete
echo "== kernel events"; cat gpurun_out/kernel_events.txt 2>/dev/null | head -50
echo "== trace reports"; tail -n 8 gpurun_out/trace_*.txt 2>/dev/null
