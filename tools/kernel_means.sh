#!/bin/bash
# Mean duration per kernel instantiation over a 20-step run of the default bench (rocprofv3 kernel trace),
# steady-state steps only (the first 30 % of each kernel's launches are dropped).
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
rm -rf $R/gpurun_out/prof_k
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_k -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing "$@" > $R/gpurun_out/prof_k.log 2>&1 || { tail -5 $R/gpurun_out/prof_k.log; exit 1; }
python3 - <<PY
import csv, glob, re
from collections import defaultdict
f = sorted(glob.glob("$R/gpurun_out/prof_k/*/*kernel_trace.csv"))[-1]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
total = 0
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "bn::" not in name: continue
    tail = v[len(v) * 3 // 10:]
    print("%-60s launches %4d  mean %8.1f us" % (name[:60], len(v), sum(tail) / len(tail)))
PY
tail -c 200 $R/gpurun_out/prof_k.log
