#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/h_tests.log 2>&1; echo "tests exit $?"
tail -5 gpurun_out/r3/h_tests.log
timeout -k 10 300 python tools/r3_hybrid_ab.py 512 > gpurun_out/r3/hybrid_ab.log 2>&1; echo "hybrid ab exit $?"
cat gpurun_out/r3/hybrid_ab.log
timeout -k 10 300 python tools/r3_hybrid_ab.py 512 side > gpurun_out/r3/hybrid_ab_side.log 2>&1; echo "hybrid ab side exit $?"
cat gpurun_out/r3/hybrid_ab_side.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/hybrid_prof -- python3 $R/tools/hybrid_prof.py > $R/gpurun_out/r3/hybrid_prof.log 2>&1); echo "hybrid prof exit $?"
find gpurun_out/r3/hybrid_prof -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-220
find gpurun_out/r3 -size +8M -delete 2>/dev/null
