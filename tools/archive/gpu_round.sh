#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats + PMC passes.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json
if [ "$1" = "prof" ]; then
  rm -rf gpurun_out/prof
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1); echo "rocprof stats exit $?"
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1); echo "rocprof fetch exit $?"
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1); echo "rocprof write exit $?"
  find gpurun_out/prof -name "*.csv" | head -30
  # keep only small summaries (kernel_stats) + the counter csvs; drop giant traces if any
  find gpurun_out/prof -size +8M -delete
fi
