#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && PDEGPU_DUMP_MAPS=$R/gpurun_out/r3/c2p_maps.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/c2p -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r3/c2p.log 2>&1); echo "c2 persistent profile exit $?"
timeout -k 10 900 tools/r3_placement2.sh > /dev/null; echo "placement2 exit $?"
cat gpurun_out/r3/placement2.log
