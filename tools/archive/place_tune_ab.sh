mkdir -p gpurun_out/s3; for i in 1 2 3 4 5; do for pt in 0 1; do PDEGPU_PLACE_TUNE=$pt PDEGPU_DEBUG_GEOM=1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>gpurun_out/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('place_tune=$pt step', round(d['ms_per_step'],4), 'K1', round(r['avg_ms'],4), 'K2', round(r['other_kernels'][0]['avg_ms'],4), 'native', round(d['native_loop']['ms_per_step'],4))
"; grep "pdegpu placement" gpurun_out/err.txt | head -1 | cut -c1-140; done; done > gpurun_out/s3/place_tune.log 2>&1; cat gpurun_out/s3/place_tune.log
