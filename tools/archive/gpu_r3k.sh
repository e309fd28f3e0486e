#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python tools/r3_el_sizes.py > gpurun_out/r3/el_sizes2.log 2>&1; echo "el sizes exit $?"
cat gpurun_out/r3/el_sizes2.log
