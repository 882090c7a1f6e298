for m in gemnet escn equiformer; do
timeout 400 python scripts/bench_$m.py --molecules 16 --steps 4 --warmup 2 --kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
tf=d['gemm_classes_TFLOPs']; ks=d['kernel_ms_per_step']
print('$m', round(d['ms_per_step'],1), 'gemm', round(d.get('gemm_ms_per_step', d['roofline'].get('gemm_ms_per_step',0)),1), 'frac', round(d['roofline']['frac'],3))
rows=[(k,ks[k][0],ks[k][1],tf[k]) for k in ks if k in tf]
for r in sorted(rows,key=lambda r:-r[1]): print('   %-30s %7.3f ms %4d x %6.1f TF %7.1f us'%(r[0],r[1],r[2],r[3],1e3*r[1]/r[2]))
"
done
