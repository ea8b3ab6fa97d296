#!/bin/bash
# Kernel trace of float64 inference on the small fixture networks at 2^20 observations.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
rm -rf $R/gpurun_out/prof_f64
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f64 -- python3 $R/tools/f64_small_networks.py > $R/gpurun_out/prof_f64.log 2>&1 || { tail -5 $R/gpurun_out/prof_f64.log; exit 1; }
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/prof_f64/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-90s calls %5s avg %10.1f us  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
