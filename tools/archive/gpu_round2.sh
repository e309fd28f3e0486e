#!/bin/bash
# Round-2 profile set on one GPU box: bench.py, rocprofv3 kernel stats + FETCH/WRITE PMC passes of the same command, the config 2 / 5
# kernels, the elasticity apply counters, the streaming ceilings and the per-rank slab surrogate.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 600 python bench.py > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err; echo "bench exit $?"
cat gpurun_out/r2p/bench.json | cut -c1-600
rm -rf gpurun_out/prof
B="python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B > $R/gpurun_out/r2p/prof_stats.log 2>&1); echo "rocprof stats exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2p/prof_fetch.log 2>&1); echo "rocprof fetch exit $?"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2p/prof_write.log 2>&1); echo "rocprof write exit $?"
# elasticity apply: kernel stats + counters (config 5 and 512x128x128)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_el/stats -- python3 $R/tools/el_prof.py > $R/gpurun_out/r2p/prof_el_stats.log 2>&1); echo "el stats exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_el/fetch -- python3 $R/tools/el_prof.py > $R/gpurun_out/r2p/prof_el_fetch.log 2>&1); echo "el fetch exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_el/write -- python3 $R/tools/el_prof.py > $R/gpurun_out/r2p/prof_el_write.log 2>&1); echo "el write exit $?"
# configs 2 and 5 end to end (launch-per-kernel 2-D loop: the cooperative kernel is profiled separately below)
(cd /tmp && PDEGPU_PERSIST2D=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c25/stats -- python3 $R/tools/c2_prof.py > $R/gpurun_out/r2p/prof_c25.log 2>&1); echo "c25 stats exit $?"
# (the cooperative persistent 2-D kernel is NOT profiled: rocprofv3 7.2 aborts on cooperative launches, profiles/r02_rocprofv3_cooperative_launch_abort.log)
for t in membw marchbw; do [ -x tools/build/$t ] || { mkdir -p tools/build; /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/$t.hip -o tools/build/$t; }; done
timeout -k 10 300 tools/build/membw > gpurun_out/r2p/membw.log 2>&1; echo "membw exit $?"
timeout -k 10 200 tools/build/marchbw > gpurun_out/r2p/marchbw.log 2>&1; echo "marchbw exit $?"
timeout -k 10 300 python tools/slab_surrogate.py gpurun_out/r2p/slab_surrogate.json > gpurun_out/r2p/surrogate.log 2>&1; echo "surrogate exit $?"
timeout -k 10 200 python tools/c2_persist.py 40 > gpurun_out/r2p/c2_persist.log 2>&1; echo "c2_persist exit $?"
timeout -k 10 200 python tools/hybrid_bench.py 512 > gpurun_out/r2p/hybrid.log 2>&1; echo "hybrid exit $?"
timeout -k 10 200 python tools/r2_exp12.py > gpurun_out/r2p/exp12.log 2>&1; echo "exp12 exit $?"
timeout -k 10 300 python tools/r2_exp13.py > gpurun_out/r2p/exp13.log 2>&1; echo "exp13 exit $?"
timeout -k 10 300 python tools/r2_exp13.py 512 128 128 >> gpurun_out/r2p/exp13.log 2>&1; echo "exp13 large exit $?"
find gpurun_out/prof gpurun_out/prof_el gpurun_out/prof_c25 gpurun_out/prof_c2p -size +8M -delete 2>/dev/null
find gpurun_out/prof* -name "*kernel_stats.csv" | head
