#!/bin/bash
# single-reduction slab CG: parity tests of the new paths, per-rank surrogate (both variants), rank-8 launch trace
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python -m pytest tests/test_gpu_heat.py tests/test_gpu_multirank.py -m gpu -x -q -k "slabs_on_one_gpu or multi_process or bench_line" > gpurun_out/r3/sr_tests.log 2>&1; echo "sr tests exit $?"
tail -5 gpurun_out/r3/sr_tests.log
timeout -k 10 400 python tools/slab_surrogate.py gpurun_out/r3/slab_surrogate.json > gpurun_out/r3/surrogate.log 2>&1; echo "surrogate exit $?"
grep "'world': 8\|'world': 1," gpurun_out/r3/surrogate.log | cut -c1-330
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/rank8sr -- python3 $R/tools/r3_rank_trace.py 8 sr > $R/gpurun_out/r3/rank8sr.log 2>&1); echo "rank8 sr trace exit $?"
find gpurun_out/r3 -size +8M -delete 2>/dev/null
