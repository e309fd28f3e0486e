#!/bin/bash
# A/B of a tunable on the weighted kernels by kernel time: rocprofv3 stats of tools/w_prof.py $1 under each setting given as further arguments
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
N=$1; shift
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr -c 'A-Za-z0-9\n' '_')
  rm -rf $R/gpurun_out/prof_ab/$tag
  (cd /tmp && env $cfg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab/$tag -- python3 $R/tools/w_prof.py $N > /dev/null 2>&1)
  echo "== $cfg (exit $?)"
  find $R/gpurun_out/prof_ab/$tag -name "*kernel_stats.csv" -exec grep "k_wmarch\|k_wk2" {} \; | awk -F'",' '{print $1}' | cut -c1-60 | paste - <(find $R/gpurun_out/prof_ab/$tag -name "*kernel_stats.csv" -exec grep "k_wmarch\|k_wk2" {} \; | awk -F'",' '{print $2}' | awk -F, '{print $1, $3}') | head -6
  find $R/gpurun_out/prof_ab/$tag -name "*_kernel_trace.csv" -delete
done
