#!/bin/bash
# deferred x update of the elasticity PCG (el_xdefer): parity, PCG per iteration with / without, kernel averages under rocprofv3
python -m pytest tests/test_gpu_elasticity.py tests/test_gpu_fullsize.py -x -q -k "elasticity or config5 or deferred or tiny" > gpurun_out/el_tests.log 2>&1; tail -4 gpurun_out/el_tests.log
for cfg in "PDEGPU_EL_XDEFER=1" "PDEGPU_EL_XDEFER=0" "PDEGPU_EL_XDEFER=1" "PDEGPU_EL_XDEFER=0"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python tools/r3_el_iter.py 2>&1 | grep "variant 1"
done | tee gpurun_out/el_xdefer.log
export TMPDIR=/tmp; R=$(pwd)
(cd /tmp && PDEGPU_PERSIST2D_LAUNCH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_xd/stats -- python3 $R/tools/c2_prof.py > $R/gpurun_out/prof_xd.log 2>&1); echo "rocprof exit $?"
find gpurun_out/prof_xd -name "*_kernel_trace.csv" -delete
head -4 $(ls -t gpurun_out/prof_xd/stats/*/*kernel_stats.csv | head -1) | cut -c1-200
