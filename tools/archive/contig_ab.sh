for i in 1 2 3 4; do for vc in 0 1; do PDEGPU_VEC_CONTIG=$vc PDEGPU_PLACE_TUNE=0 PDEGPU_DEBUG_GEOM=1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>gpurun_out/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('vec_contig=$vc step', round(d['ms_per_step'],4), 'K1', round(r['avg_ms'],4), 'K2', round(r['other_kernels'][0]['avg_ms'],4), 'spmv', round(r['spmv_only']['avg_ms'],4), 'native', round(d['native_loop']['ms_per_step'],4))
"; grep "pdegpu alloc" gpurun_out/err.txt | sort | uniq -c | tr "\n" ";"; echo; done; done
