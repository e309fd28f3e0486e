#!/bin/bash
# weighted marching kernels: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs) of one spherical and one composite-core solve
# at $1^3 cells (default 192) -> gpurun_out/prof_w$1/{stats,fetch,write}; tools/summarize_profiles.py r04 weighted_$1 prof_w$1 condenses them
set -o pipefail
N=${1:-192}
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf gpurun_out/prof_w$N
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w$N/stats -- python3 $R/tools/w_prof.py $N > $R/gpurun_out/r4/prof_w$N.log 2>&1); echo "rocprof stats exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_w$N/fetch -- python3 $R/tools/w_prof.py $N > $R/gpurun_out/r4/prof_w${N}_fetch.log 2>&1); echo "rocprof fetch exit $?"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_w$N/write -- python3 $R/tools/w_prof.py $N > $R/gpurun_out/r4/prof_w${N}_write.log 2>&1); echo "rocprof write exit $?"
cat gpurun_out/r4/prof_w$N.log | grep -v amdgpu.ids
find gpurun_out/prof_w$N -name "*_kernel_trace.csv" -delete 2>/dev/null
python3 - <<PY
import csv, glob, collections
for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof_w$N/%s/*/*_counter_collection.csv" % kind):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
        # keep only a compact per-kernel table: the raw collection is large
        with open(f + ".summary.txt", "w") as o:
            for k, v in acc.items():
                v.sort()
                o.write("%s\t%d\t%.1f\n" % (k, len(v), v[len(v) // 2]))
    for k, v in acc.items():
        v.sort()
        if v[len(v) // 2] > 1024:
            print(name, k[:80], len(v), "median KiB", v[len(v) // 2])
PY
