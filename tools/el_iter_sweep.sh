#!/bin/bash
# elasticity PCG per-iteration time (config 5 and 512 x 128 x 128) under a few tunables: workgroups of the flat passes (el_grid_mult; 9 = one node pair
# per thread at config 5) with the block sums in the update pass's prologue (lazy_dot = 1) or in a reduce launch of their own (lazy_dot = 0)
for cfg in "" "PDEGPU_LAZY_DOT=0" "PDEGPU_LAZY_DOT=0 PDEGPU_EL_GRID_MULT=9" "PDEGPU_LAZY_DOT=0 PDEGPU_EL_GRID_MULT=18" "PDEGPU_LAZY_DOT=0 PDEGPU_EL_GRID_MULT=6" "PDEGPU_EL_GRID_MULT=9"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/r3_el_iter.py 2>&1 | grep "variant 1"
done
