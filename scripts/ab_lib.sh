for lib in nabladft_amd/_ablate/libnablaq_head.so nabladft_amd/libnablaq.so; do
  echo "== $lib"
  NABLAQ_LIB=$PWD/$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('msgf') or k.startswith('gwr')}, d['ms_per_step'])"
done
