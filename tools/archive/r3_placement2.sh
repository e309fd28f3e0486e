#!/bin/bash
# second allocation experiment: one physically contiguous arena, vectors staggered; contiguous per-vector allocations with the placement tuner
mkdir -p gpurun_out/r3
L=gpurun_out/r3/placement2.log
for round in 1 2 3; do
  PDEGPU_VEC_CONTIG=0 timeout -k 10 120 python tools/r3_placement.py >> $L 2>> gpurun_out/r3/placement2.err
  PDEGPU_VEC_CONTIG=1 timeout -k 10 120 python tools/r3_placement.py >> $L 2>> gpurun_out/r3/placement2.err
  PDEGPU_VEC_CONTIG=1 PDEGPU_PLACE_TUNE=1 timeout -k 10 120 python tools/r3_placement.py >> $L 2>> gpurun_out/r3/placement2.err
  for st in 0 4 64 132 260 516 1028 2052 4100 8196 16388 65540; do
    PDEGPU_VEC_CONTIG=6 PDEGPU_VEC_STAGGER_KB=$st timeout -k 10 120 python tools/r3_placement.py >> $L 2>> gpurun_out/r3/placement2.err || echo "mode 6 stagger $st failed" >> $L
  done
done
cat $L
