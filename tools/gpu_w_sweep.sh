#!/bin/bash
# weighted marching kernels: rocprofv3 kernel stats of one spherical + one composite-core solve at 192^3, then the rates under a few tunables
set -o pipefail
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf gpurun_out/prof_w
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w/stats -- python3 $R/tools/w_prof.py 192 > $R/gpurun_out/r4/prof_w.log 2>&1); echo "rocprof stats exit $?"
find gpurun_out/prof_w -name "*kernel_stats.csv" -exec head -12 {} \;
find gpurun_out/prof_w -size +8M -name "*_kernel_trace.csv" -delete 2>/dev/null
for cfg in "" "PDEGPU_WFAST_RY=1" "PDEGPU_WFAST_WG_PER_CU=2" "PDEGPU_WFAST_WG_PER_CU=3" "PDEGPU_WFAST_WG_PER_CU=4" "PDEGPU_WFAST_WG_PER_CU=6" "PDEGPU_WFAST_RY=1 PDEGPU_WFAST_WG_PER_CU=2" "PDEGPU_WFAST_K2_MULT=2" "PDEGPU_WFAST_K2_MULT=8"; do
  echo "== $cfg"
  env $cfg PDEGPU_DEBUG_GEOM=1 timeout -k 10 200 python tools/w_rate.py 192 2>&1 | grep -v "plain box\|amdgpu.ids" | grep "192\|wmarch"
done
