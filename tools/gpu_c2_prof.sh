#!/bin/bash
# configs 2 and 5 end to end under rocprofv3 --kernel-trace --stats, the persistent 2-D kernel launched as an ordinary kernel (tools/c2_prof.py sets
# persist2d_launch = 1): the run must exit 0
set -o pipefail
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf gpurun_out/prof_c25
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c25/stats -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r4/prof_c25.log 2>&1); echo "c25 stats exit $?"
grep -v "amdgpu.ids\|rocprofv3\]" gpurun_out/r4/prof_c25.log | tail -5
find gpurun_out/prof_c25 -name "*_kernel_trace.csv" -delete 2>/dev/null
find gpurun_out/prof_c25 -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-200
