#!/bin/bash
# Matrix-core utilisation of the dense factor / EM kernels on the deep-tree model:
# one rocprofv3 --pmc pass (2 SQ counters + 1 GRBM counter).  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_mfma
rm -rf $OUT
mkdir -p $OUT
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py --model c5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/run.log 2>&1 || { echo "pass failed"; tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, json, re
from collections import defaultdict
path = glob.glob("$OUT/*/*counter_collection.csv")[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    if "mfma" in r["Kernel_Name"]:
        acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    insts = sum(c["SQ_INSTS_VALU_MFMA_F32"]) / len(c["SQ_INSTS_VALU_MFMA_F32"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    # MfmaUtil as rocprofv3 derives it: busy cycles summed over SIMDs / (active cycles * number of SIMDs)
    out[k] = dict(launches=len(c["GRBM_GUI_ACTIVE"]), mfma_busy_cycles=busy, mfma_insts_f32=insts, grbm_gui_active=gui,
                  mfma_util_percent=100.0 * busy / (gui / 8.0 * 1024.0))
    print(k, out[k])
json.dump(out, open("$R/gpurun_out/pmc_mfma.json", "w"), indent=1)
PY
