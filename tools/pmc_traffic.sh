#!/bin/bash
# HBM traffic per kernel launch from the L2 memory-side counters, one rocprofv3 --pmc pass per
# counter (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root:
#   [PMC_TAG=_c3b] bash tools/pmc_traffic.sh [extra bench.py args, e.g. --model c3b]
# Writes gpurun_out/pmc_traffic$PMC_TAG.json; copy it to profiles/ to have bench.py report `traffic`.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${PMC_TAG:-}
OUT=$R/gpurun_out/pmc$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$C
  timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing "$@" > $OUT/$C.log 2>&1 || { echo "pass $C failed"; tail -5 $OUT/$C.log; exit 1; }
done
python3 $R/tools/pmc_summarise.py $OUT $R/gpurun_out/pmc_traffic$TAG.json
