#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python tools/r3_hybrid_ab.py 512 > gpurun_out/r3/hybrid_ab2.log 2>&1; echo "hybrid ab exit $?"
cat gpurun_out/r3/hybrid_ab2.log
for sz in "641 161 161" "257 257 257" "513 129 129"; do timeout -k 10 300 tools/build/marchbw $sz > gpurun_out/r3/marchbw_$(echo $sz | tr ' ' 'x').log 2>&1; echo "marchbw $sz exit $?"; done
cat gpurun_out/r3/marchbw_641x161x161.log
