"""Condense gpurun_out/prof (rocprofv3 --kernel-trace --stats and separate --pmc FETCH_SIZE / WRITE_SIZE passes)
into the small, committed files under profiles/.   usage: python tools/summarize_profiles.py r01 [tag [profile-folder under gpurun_out]]

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so the
read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  (Calibration inside this very
profile: k_k2 reads two full vectors and writes one -> corrected fetch = 2.000 x write.)"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
tag = ("_" + sys.argv[2]) if len(sys.argv) > 2 else ""
prof = ROOT / "gpurun_out" / (sys.argv[3] if len(sys.argv) > 3 else "prof")
out = ROOT / "profiles"
out.mkdir(exist_ok=True)

import os


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)      # gpurun merges every run into the same folders
    return files[-1:] if files else []


stats = newest(str(prof / "stats" / "*" / "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], out / f"{rnd}{tag}_kernel_stats.csv")


def counters(kind, name):
    acc = collections.defaultdict(list)
    files = newest(str(prof / kind / "*" / "*_counter_collection.csv"))
    if not files:
        return acc
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == name:
            acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
summary = {"units": "bytes per launch", "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; "
           "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction); steady-state launches = median",
           "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f = sorted(fetch.get(k, [0.0]))
    w = sorted(write.get(k, [0.0]))
    fm, wm = f[len(f) // 2], w[len(w) // 2]
    if max(fm, wm) < 1024:
        continue
    summary["kernels"][k] = {"launches_sampled": len(f), "FETCH_SIZE_KiB_median": fm, "WRITE_SIZE_KiB_median": wm,
                             "hbm_read_bytes": 2 * fm * 1024, "hbm_write_bytes": wm * 1024,
                             "hbm_bytes": 2 * fm * 1024 + wm * 1024}
(out / f"{rnd}{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
import re
import subprocess
try:
    commit = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
except OSError:
    commit = "unknown"
k1 = [v for k, v in summary["kernels"].items() if re.search(r"k_fast<7, \d+, 2[,>]", k)]
if k1:
    (out / "pmc_latest.json").write_text(json.dumps({"k1_hbm_bytes_per_launch": k1[0]["hbm_bytes"], "source": f"{rnd}{tag}_pmc_summary.json", "commit": commit}))
k2 = [v for k, v in summary["kernels"].items() if re.search(r"k_fast<7, \d+, 3[,>]", k)]
if k1 and k2:
    latest = {"k1_hbm_bytes_per_launch": k1[0]["hbm_bytes"], "k2_hbm_bytes_per_launch": k2[0]["hbm_bytes"], "source": f"{rnd}{tag}_pmc_summary.json", "commit": commit}
    # sessions whose K2 carries the x update of every other iteration launch two instantiations of K2 (template XU = 0 | 1)
    k2x = [v for k, v in summary["kernels"].items() if re.search(r"k_fast<7, \d+, 3, \d+, \d+, \d+, 1>", k)]
    k2n = [v for k, v in summary["kernels"].items() if re.search(r"k_fast<7, \d+, 3, \d+, \d+, \d+, 0>", k)]
    if k2x and k2n:
        latest["k2_hbm_bytes_per_launch"] = k2n[0]["hbm_bytes"]
        latest["k2_with_x_hbm_bytes_per_launch"] = k2x[0]["hbm_bytes"]
    (out / "pmc_latest.json").write_text(json.dumps(latest))
print(json.dumps(summary, indent=1)[:3000])
