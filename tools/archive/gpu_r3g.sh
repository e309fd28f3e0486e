#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_elasticity.py tests/test_gpu_multirank.py tests/test_boundary.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3/g_tests.log 2>&1; echo "tests exit $?"
tail -4 gpurun_out/r3/g_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err; echo "bench exit $?"
tail -3 gpurun_out/r3/bench.err | cut -c1-400
cut -c1-900 gpurun_out/r3/bench.json
timeout -k 10 400 python tools/slab_surrogate.py gpurun_out/r3/slab_surrogate.json > gpurun_out/r3/surrogate.log 2>&1; echo "surrogate exit $?"
grep "elasticity" gpurun_out/r3/surrogate.log | cut -c1-300
