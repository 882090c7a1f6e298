timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "gemm" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_qhnet_gpu.py tests/test_graphed_gpu.py -x -q 2>&1 | tail -3
python bench.py --model qhnet --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('reference_batch_size_2'))"
python -c "import json; d=json.load(open('gpurun_out/bench_full.json')); print({k:v for k,v in list(d['kernel_ms_per_step'].items())[:24]})"
