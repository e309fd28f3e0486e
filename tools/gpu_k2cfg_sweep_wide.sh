mkdir -p gpurun_out/s14
for cfg in -1 306 292 308; do
  echo "== fast_cfg_k2 $cfg" >> gpurun_out/s14/k2cfg.log
  PDEGPU_FAST_CFG_K2=$cfg timeout -k 10 300 python tools/size_sweep.py 256 288 320 352 384 448 480 544 608 640 832 2>&1 | grep "^n=" >> gpurun_out/s14/k2cfg.log || exit 1
done
cut -c45-150 gpurun_out/s14/k2cfg.log
