#!/bin/bash
for cfg in "PDEGPU_WFAST_RY1=2" "PDEGPU_WFAST_RY1=1" "PDEGPU_WFAST_RY1=2 PDEGPU_WFAST_WG_PER_CU=2"; do
  echo "== $cfg"
  env $cfg PDEGPU_DEBUG_GEOM=1 timeout -k 10 300 python tools/w_rate.py 192 256 320 2>&1 | grep "spherical\|cm 1 ry . mode 2"
done
