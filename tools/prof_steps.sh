#!/bin/bash
# Kernel trace of a 20-step run; prints the period between consecutive EM steps and the busy share.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
rm -rf $R/gpurun_out/prof_s
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/prof_s.log 2>&1 || { tail -5 $R/gpurun_out/prof_s.log; exit 1; }
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/prof_s/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "m_step_rows" in r["Kernel_Name"]]
for a, b in zip(idx[-8:], idx[-7:]):
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
    gaps = sorted(((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3, rows[i]["Kernel_Name"][:40]) for i in range(a, b))[-3:]
    print("period %.1f us busy %.1f us kernels %d largest gaps %s" % (span / 1e3, busy / 1e3, b - a, gaps))
PY
tail -c 300 $R/gpurun_out/prof_s.log
