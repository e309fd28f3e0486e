#!/bin/bash
# the size sweep at a few sizes under each K2 geometry word (PDEGPU_FAST_CFG_K2 = ry + 16 wg_per_cu + 256 ex + 512 pf), one process per setting
mkdir -p gpurun_out/s13
for cfg in -1 292 306 322 802 804 274; do
  echo "== fast_cfg_k2 $cfg" >> gpurun_out/s13/k2cfg.log
  PDEGPU_FAST_CFG_K2=$cfg timeout -k 10 200 python tools/size_sweep.py 320 416 512 576 2>&1 | grep "^n=" >> gpurun_out/s13/k2cfg.log || exit 1
done
cut -c1-150 gpurun_out/s13/k2cfg.log
