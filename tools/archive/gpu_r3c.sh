#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
(cd /tmp && PDEGPU_NO_TORCH_PRELOAD=1 PDEGPU_DUMP_MAPS=$R/gpurun_out/r3/c2p_sys_maps.txt timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/c2p_sys -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r3/c2p_sys.log 2>&1); echo "c2 persistent profile, system HIP runtime only: exit $?"
tail -4 gpurun_out/r3/c2p_sys.log
grep -c "libhsa-runtime64\|libamdhip64" gpurun_out/r3/c2p_sys_maps.txt
grep "libhsa-runtime64\|libamdhip64" gpurun_out/r3/c2p_sys_maps.txt
