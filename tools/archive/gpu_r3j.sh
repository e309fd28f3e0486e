#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_elasticity.py tests/test_gpu_heat.py -m gpu -x -q -k "elasticity or hybrid or natural" > gpurun_out/r3/j_tests.log 2>&1; echo "tests exit $?"
tail -3 gpurun_out/r3/j_tests.log
timeout -k 10 600 python tools/r3_el_sizes.py > gpurun_out/r3/el_sizes.log 2>&1; echo "el sizes exit $?"
cat gpurun_out/r3/el_sizes.log
