#!/bin/bash
# Issue-side counters (one rocprofv3 --pmc pass) for any program:  bash tools/pmc_any.sh <tag> <program> [args]
# Prints per kernel: launches, VALU-busy share, cycles per VALU instruction, parked / issue-stall shares.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
PROG=$(realpath $1); shift
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT -- $PROG "$@" > $OUT/run.log 2>&1 || { echo "pass failed"; tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv, glob, json, re
from collections import defaultdict
path = glob.glob("$OUT/*/*counter_collection.csv")[0]
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if not m.get("GRBM_GUI_ACTIVE") or not m.get("SQ_WAVES"):
        continue
    o = dict(launches=len(c["GRBM_GUI_ACTIVE"]), waves=m["SQ_WAVES"],
             valu_busy=4.0 * m["SQ_ACTIVE_INST_VALU"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
             valu_insts_per_wave=m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
             valu_cycles_per_inst=4.0 * m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_INSTS_VALU"], 1),
             wave_cycles_per_wave=4.0 * m["SQ_WAVE_CYCLES"] / m["SQ_WAVES"],
             gui_cycles=m["GRBM_GUI_ACTIVE"] / 8.0,
             parked_share=m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
             issue_stall_share=m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"],
             active_share=m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"])
    out[k] = o
    print(k[:70], {a: round(b, 3) for a, b in o.items()})
json.dump(out, open("$R/gpurun_out/pmc_$TAG.json", "w"), indent=1)
PY
