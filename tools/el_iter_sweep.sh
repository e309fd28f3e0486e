#!/bin/bash
# elasticity PCG per-iteration time (config 5 and 512 x 128 x 128) under a few tunables
for cfg in "" "PDEGPU_EL_GRID_MULT=9" "PDEGPU_EL_GRID_MULT=12" "PDEGPU_EL_GRID_MULT=6" "PDEGPU_EL_GRID_MULT=2"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/r3_el_iter.py 2>&1 | grep "variant 1"
done
