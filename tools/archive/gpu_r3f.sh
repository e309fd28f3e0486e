#!/bin/bash
# full GPU suite after the single-reduction / split-elasticity changes, then the surrogate (wall without events)
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3/full_tests.log 2>&1; echo "full tests exit $?"
tail -6 gpurun_out/r3/full_tests.log
timeout -k 10 400 python tools/slab_surrogate.py gpurun_out/r3/slab_surrogate.json > gpurun_out/r3/surrogate.log 2>&1; echo "surrogate exit $?"
grep "'world': 8\|'world': 1," gpurun_out/r3/surrogate.log | cut -c1-380
