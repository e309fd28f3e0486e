mkdir -p gpurun_out/s12
for cfg in -1 290 306 322 276 308 324 804; do
  echo "== fast_cfg_k1 $cfg" >> gpurun_out/s12/k1cfg.log
  PDEGPU_FAST_CFG_K1=$cfg timeout -k 10 200 python tools/size_sweep.py 288 320 352 416 576 2>&1 | grep "^n=" >> gpurun_out/s12/k1cfg.log || exit 1
done
cat gpurun_out/s12/k1cfg.log | cut -c1-150
