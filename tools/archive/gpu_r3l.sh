#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_elasticity.py tests/test_gpu_multirank.py tests/test_gpu_fullsize.py tests/test_golden_fields.py -m gpu -x -q > gpurun_out/r3/l_tests.log 2>&1; echo "tests exit $?"
tail -12 gpurun_out/r3/l_tests.log
timeout -k 10 600 python tools/r3_el_iter.py > gpurun_out/r3/el_iter.log 2>&1; echo "el iter exit $?"
cat gpurun_out/r3/el_iter.log
