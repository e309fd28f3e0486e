#!/bin/bash
# weighted marching PCG: whole-call rates under PDEGPU_* overrides, one process per setting (each argument: "" or "KEY=V KEY2=V2"); sizes from $SIZES
# -> gpurun_out/w_sweep.log
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/w_sweep.log
if [ $# -eq 0 ]; then set -- "" "PDEGPU_WFAST_WG_PER_CU=3" "PDEGPU_WFAST_EX=1" "PDEGPU_WFAST_EX=0" "PDEGPU_WFAST_NT=0" "PDEGPU_WFAST_XDEFER=0"; fi
for s in "$@"; do
    env $s timeout -k 10 300 python tools/w_sweep.py ${SIZES:-256 384 512} 2>&1 | grep "\^3" >> gpurun_out/w_sweep.log || exit 1
done
cat gpurun_out/w_sweep.log
