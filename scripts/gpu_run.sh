#!/bin/bash
# One parameterised GPU-box recipe (round 6; replaces the per-round gpu_rNN_*.sh files, kept in git history).  Usage, as the gpurun command:
#   bash scripts/gpu_run.sh <out-subdir> <step> [<step> ...]
# steps:  tests[:<pytest -k expression>]   GPU suite (or a selection) -> pytest.log
#         smoke                            __graft_entry__.smoke()
#         bench[:<extra bench.py args>]    default driver command -> bench_default.json, full record, kernel events
#         benchq                           bench.py --model qhnet
#         full                             bench.py --full
#         variants[:<bench args>]          every nabladft_amd/_variants/libnablaq_*.so next to the shipped library (scripts/variants.sh build ...)
#         prof:<name>:<command...>         rocprofv3 --kernel-trace --stats of a command -> <name>_kernel_stats.csv
#         pmc:<B>:<command...>             FETCH_SIZE / WRITE_SIZE passes (separate runs) -> pmc_traffic_<B>.json
#         sq:<B>:<kernel filter>           three SQ counter passes (wave cycles / waits, instruction counts, LDS + matrix-core busy) of bench.py at batch B -> pmc_sq_<filter>.txt
#         sweep[:<models>]                 scripts/batch_sweep.py (throughput vs conformers per step of QHNet / GemNet-OC / eSCN / EquiformerV2) -> batch_sweep.json
#         sh:<command...>                  anything else
OUT=gpurun_out/$1; shift; mkdir -p $OUT; export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  echo "== $step"
  case $kind in
    tests) if [ -n "$arg" ]; then timeout -k 5 1500 python -m pytest tests -q -m gpu -x -k "$arg" 2>&1 | tail -25 | tee $OUT/pytest.log; else timeout -k 5 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest.log; fi
           cp gpurun_out/parity_report.txt gpurun_out/mixed_sizes_report.txt $OUT/ 2>/dev/null ;;
    smoke) timeout -k 5 600 python __graft_entry__.py --smoke 2>&1 | tail -8 | tee $OUT/smoke.log ;;
    bench) S=$(date +%s); timeout -k 5 700 python bench.py $arg > $OUT/bench_default.stdout 2> $OUT/bench_default.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_default.wall
           tail -1 $OUT/bench_default.stdout > $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_default_full_record.json 2>/dev/null; cp gpurun_out/kernel_events.txt $OUT/kernel_events.txt 2>/dev/null
           python - $OUT/bench_default.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d[k] for k in ("value", "ms_per_step")}, {k: round(v, 3) for k, v in sorted(d.get("kernel_ms_per_step", {}).items(), key=lambda kv: -kv[1])[:14]})
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_launch_ms")}, "mae", d.get("mae_vs_cpu_reference"))
except Exception as e:
    print("no bench record:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
           ;;
    benchq) timeout -k 5 500 python bench.py --model qhnet > $OUT/bench_qhnet.stdout 2> $OUT/bench_qhnet.err; tail -1 $OUT/bench_qhnet.stdout > $OUT/bench_qhnet.json; cp gpurun_out/bench_full.json $OUT/bench_qhnet_full_record.json; tail -c 600 $OUT/bench_qhnet.json; echo ;;
    full) S=$(date +%s); timeout -k 5 1500 python bench.py --full > $OUT/bench_full.stdout 2> $OUT/bench_full.err; echo "wall $(( $(date +%s) - S )) s" | tee $OUT/bench_full.wall; cp gpurun_out/bench_full.json $OUT/bench_full_record.json ;;
    variants) for lib in nabladft_amd/libnablaq.so nabladft_amd/_variants/libnablaq_*.so; do
                echo "-- $lib"
                NABLAQ_LIB=$PWD/$lib timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $arg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print({n:round(k.get(n,0),3) for n in '${VAR_KEYS:-gwr_mol,msgf_rev_dual_ng,msgf_tan,msgf_rev_force,msgf_fwd,upd_rev}'.split(',')}, round(sum(k.values()),3), round(d['ms_per_step'],3))"
              done 2>&1 | tee $OUT/variants.txt ;;
    prof) name=${arg%%:*}; cmd=${arg#*:}; rm -rf $OUT/prof_$name
          timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- $cmd > $OUT/rocprof_$name.log 2>&1
          f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && head -12 "$f"; rm -rf $OUT/prof_$name ;;
    pmc) B=${arg%%:*}; cmd=${arg#*:}; rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
         timeout -k 5 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- $cmd > $OUT/pmc_fetch.log 2>&1
         timeout -k 5 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- $cmd > $OUT/pmc_write.log 2>&1
         python scripts/pmc_summary.py $B "$cmd" $OUT/pmc_traffic_$B.json | head -16 | tee $OUT/pmc_traffic_$B.txt; rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write ;;
    sq) B=${arg%%:*}; FILT=${arg#*:}
        sqrun() { tag=$1; shift; rm -rf gpurun_out/pmc_$tag
          timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -o p -- python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-roofline > $OUT/pmc_$tag.log 2>&1
          python scripts/pmc_sq_summary.py $tag "$FILT"; rm -rf gpurun_out/pmc_$tag; }
        { sqrun sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES
          sqrun sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
          sqrun sq3 SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM; } 2>&1 | tee $OUT/pmc_sq_${FILT//,/_}.txt ;;
    sweep) timeout -k 5 2400 python scripts/batch_sweep.py --out $OUT/batch_sweep.json ${arg:+--models $arg} 2>&1 | tail -30 | tee $OUT/batch_sweep.log ;;
    sh) bash -c "$arg" 2>&1 | tail -40 | tee -a $OUT/sh.log ;;
    *) echo "unknown step $kind" ;;
  esac
done
