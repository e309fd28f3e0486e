#!/bin/bash
# issue-slot counters of the marching elasticity apply (config 5 and 512 x 128 x 128): is it bound by VALU issue?
R=$(pwd); export TMPDIR=/tmp
mkdir -p $R/gpurun_out/elpmc
cd /tmp
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_')
  PDEGPU_EL_FAST_XH=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/elpmc/$tag -- python3 $R/tools/el_prof.py > $R/gpurun_out/elpmc/$tag.log 2>&1
  echo "pmc $tag exit $?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/elpmc/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if 'k_el_fast' not in k and 'k_el_xh' not in k: continue
        agg[(k[:60], row['Grid_Size'])][row['Counter_Name']].append(float(row['Counter_Value']))
    for (k, gsz), d in agg.items():
        print(k, 'grid', gsz, {c: sum(v) / len(v) for c, v in d.items()})
PY
