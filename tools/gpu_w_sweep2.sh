#!/bin/bash
for cfg in "" "PDEGPU_WFAST_K2_NT=0" "PDEGPU_WFAST_K2_NT=1" "PDEGPU_WFAST_K2_MULT=2" "PDEGPU_WFAST_K2_MULT=8" "PDEGPU_WFAST_WG_PER_CU=2" "PDEGPU_WFAST_WG_PER_CU=4"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python tools/w_rate.py 192 256 320 2>&1 | grep -v "plain box\|amdgpu.ids" | grep "\^3"
done
