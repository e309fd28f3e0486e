#!/bin/bash
# round 4, second session: one-round refit of the marching elasticity apply (el_fast_fit) — parity, per-workgroup timelines, PCG per iteration
python -m pytest tests/test_gpu_elasticity.py -x -q > gpurun_out/el_tests.log 2>&1; tail -5 gpurun_out/el_tests.log
for s in "el_fast_fit=1" "el_fast_fit=0" "el_fast_fit=1 el_bnd_gran=256" "el_fast_fit=0 el_bnd_gran=256"; do
  echo "== $s"; python tools/el_trace.py $s 2>&1 | grep -v amdgpu.ids
done > gpurun_out/el_trace6.log 2>&1
cat gpurun_out/el_trace6.log
for cfg in "PDEGPU_EL_FAST_FIT=1" "PDEGPU_EL_FAST_FIT=0" "PDEGPU_EL_FAST_FIT=1 PDEGPU_EL_BND_GRAN=256" "PDEGPU_EL_FAST_FIT=0 PDEGPU_EL_BND_GRAN=256"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python tools/r3_el_iter.py 2>&1 | grep "variant 1"
done | tee gpurun_out/el_iter.log
