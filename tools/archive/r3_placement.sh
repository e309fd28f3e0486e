#!/bin/bash
# alternating fresh processes over the allocation modes (verdict item 2); output: gpurun_out/r3/placement.log
mkdir -p gpurun_out/r3
for round in 1 2 3 4 5; do
  for m in 0 1 2 3 4 5; do
    PDEGPU_VEC_CONTIG=$m PDEGPU_DEBUG_GEOM=1 timeout -k 10 120 python tools/r3_placement.py >> gpurun_out/r3/placement.log 2>> gpurun_out/r3/placement.err || echo "mode $m failed" >> gpurun_out/r3/placement.log
  done
done
cat gpurun_out/r3/placement.log
