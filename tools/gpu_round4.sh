#!/bin/bash
# Round-4 profile set on one GPU box: bench.py, rocprofv3 kernel stats + FETCH/WRITE PMC passes of the same command (separate runs), configs 2 / 5 end
# to end (persistent 2-D kernel as an ordinary launch: rocprofv3 exits 0), the weighted kernels at 192^3 and 256^3 (stats + PMC), weighted rates by size,
# the size sweep of the heat CG kernels.  Outputs under gpurun_out/; tools/summarize_profiles.py condenses them into profiles/.
set -o pipefail
mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err; echo "bench exit $?"
cut -c1-400 gpurun_out/r4p/bench.json
rm -rf gpurun_out/prof
B="python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B > $R/gpurun_out/r4p/prof_stats.log 2>&1); echo "rocprof stats exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r4p/prof_fetch.log 2>&1); echo "rocprof fetch exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r4p/prof_write.log 2>&1); echo "rocprof write exit $?"
grep "^{" gpurun_out/r4p/prof_stats.log > gpurun_out/r4p/bench_profiled_run.json
find gpurun_out/prof -name "*_kernel_trace.csv" -delete 2>/dev/null
bash tools/gpu_c2_prof.sh > gpurun_out/r4p/c25.log 2>&1; tail -12 gpurun_out/r4p/c25.log | cut -c1-200
bash tools/gpu_w_prof.sh 192 > gpurun_out/r4p/w192.log 2>&1; grep "exit" gpurun_out/r4p/w192.log
bash tools/gpu_w_prof.sh 256 > gpurun_out/r4p/w256.log 2>&1; grep "exit" gpurun_out/r4p/w256.log
timeout -k 10 600 python tools/w_rate.py 128 192 256 320 384 512 > gpurun_out/r4p/w_rate.log 2>&1; echo "w_rate exit $?"; grep "\^3" gpurun_out/r4p/w_rate.log | cut -c1-200
timeout -k 10 300 python tools/r3_el_iter.py > gpurun_out/r4p/el_iter.log 2>&1; echo "el iter exit $?"; grep variant gpurun_out/r4p/el_iter.log
timeout -k 10 600 python tools/size_sweep.py 128 160 192 224 256 288 320 352 384 400 448 480 512 544 576 640 700 768 896 1024 > gpurun_out/r4p/size_sweep.log 2>&1; echo "size sweep exit $?"
find gpurun_out/prof* -name "*kernel_stats.csv" | head
