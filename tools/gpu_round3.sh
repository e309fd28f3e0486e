#!/bin/bash
# Round-3 profile set on one GPU box: bench.py, rocprofv3 kernel stats + FETCH/WRITE PMC passes of the same command, configs 2 / 5 end to end,
# the natural-face iteration, the per-rank surrogate (both CG variants) and the launch trace of one rank of 8.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out/r3p
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python bench.py > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err; echo "bench exit $?"
cut -c1-500 gpurun_out/r3p/bench.json
rm -rf gpurun_out/prof
B="python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B > $R/gpurun_out/r3p/prof_stats.log 2>&1); echo "rocprof stats exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r3p/prof_fetch.log 2>&1); echo "rocprof fetch exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r3p/prof_write.log 2>&1); echo "rocprof write exit $?"
# configs 2 and 5 end to end, default paths (the cooperative 2-D kernel: the profiler faults at exit AFTER writing its files, profiles/r03_rocprofv3_exit_abort.log)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c25/stats -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r3p/prof_c25.log 2>&1); echo "c25 stats exit $? (139 expected)"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_hyb/stats -- python3 $R/tools/hybrid_prof.py > $R/gpurun_out/r3p/prof_hyb.log 2>&1); echo "hybrid stats exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rank8/stats -- python3 $R/tools/r3_rank_trace.py 8 sr > $R/gpurun_out/r3p/prof_rank8.log 2>&1); echo "rank8 stats exit $?"
timeout -k 10 400 python tools/slab_surrogate.py gpurun_out/r3p/slab_surrogate.json > gpurun_out/r3p/surrogate.log 2>&1; echo "surrogate exit $?"
timeout -k 10 300 python tools/r3_el_iter.py > gpurun_out/r3p/el_iter.log 2>&1; echo "el iter exit $?"
timeout -k 10 300 python tools/hybrid_bench.py 512 > gpurun_out/r3p/hybrid.log 2>&1; echo "hybrid exit $?"
find gpurun_out/prof gpurun_out/prof_c25 gpurun_out/prof_hyb gpurun_out/prof_rank8 -size +8M -name "*_kernel_trace.csv" -delete 2>/dev/null
find gpurun_out/prof* -name "*kernel_stats.csv" | head
