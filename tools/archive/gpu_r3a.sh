#!/bin/bash
# round-3 first GPU batch: per-rank launch trace (N = 8 surrogate), config-2 profile with the cooperative kernel and explicit teardown,
# allocation-mode experiment
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/rank8 -- python3 $R/tools/r3_rank_trace.py 8 > $R/gpurun_out/r3/rank8.log 2>&1); echo "rank8 trace exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/c2p -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r3/c2p.log 2>&1); echo "c2 persistent profile exit $?"
tail -5 gpurun_out/r3/c2p.log
timeout -k 10 900 tools/r3_placement.sh > /dev/null; echo "placement exit $?"
cat gpurun_out/r3/placement.log
find gpurun_out/r3 -size +8M -delete 2>/dev/null
find gpurun_out/r3 -name "*kernel_stats.csv"
