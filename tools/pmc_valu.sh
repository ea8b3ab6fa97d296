#!/bin/bash
# Issue-side picture of the kernels of one EM step: one rocprofv3 --pmc pass with seven SQ counters
# and GRBM_GUI_ACTIVE.  Run on the GPU box from the repo root:  [PMC_TAG=_c3b] bash tools/pmc_valu.sh [bench args]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${PMC_TAG:-}
OUT=$R/gpurun_out/pmc_valu$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing "$@" > $OUT/run.log 2>&1 || { echo "pass failed"; tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, json, re
from collections import defaultdict
path = glob.glob("$OUT/*/*counter_collection.csv")[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if name.startswith("bn::"):
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["launches"] = len(c["GRBM_GUI_ACTIVE"])
    # Share of the launch's SIMD cycles in which a vector instruction executes.  SQ_ACTIVE_INST_VALU
    # counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is the launch's active
    # cycles summed over the 8 XCDs.
    m["valu_busy"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if m["GRBM_GUI_ACTIVE"] else None
    m["valu_cycles_per_inst"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"] if m["SQ_INSTS_VALU"] else None
    m["parked_share_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else None
    m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"] if m["SQ_WAVES"] else None
    m["wait_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"] if m["SQ_WAVE_CYCLES"] else None
    out[k] = m
    print(k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in m.items()})
json.dump(out, open("$R/gpurun_out/pmc_valu$TAG.json", "w"), indent=1)
PY
