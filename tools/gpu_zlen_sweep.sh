#!/bin/bash
# K1 / K2 of the steady heat path at row widths away from 128 * 2^k + 1, by forced z-chunk length (0 = the fitted policy); sizes from $SIZES
for z in ${ZLENS:-0 12 16 24 32 48 64 96 144 288}; do
  echo "== fast_zlen $z"
  PDEGPU_FAST_ZLEN=$z PDEGPU_DEBUG_GEOM=1 timeout -k 10 200 python tools/size_sweep.py ${SIZES:-352 544 576} 2>&1 | grep "n=\|mode 2 npt" | sed 's/\[pdegpu geom\] //' | cut -c1-200
done
