#!/bin/bash
# Everything profiles/ holds for one round, from the current tree, on the GPU box:
#   bash tools/refresh_profiles.sh r1
# Leaves the files under gpurun_out/profiles_<tag>/; copy them into profiles/ afterwards.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
# 1. the default bench line (with roofline and cpu_baseline), plus the other two models
timeout -k 10 300 python3 bench.py > $OUT/${TAG}_bench_c3a_60k.json 2> $OUT/bench_c3a.err || { echo "bench failed"; tail -5 $OUT/bench_c3a.err; exit 1; }
timeout -k 10 300 python3 bench.py --model c3b --no-cpu-baseline > $OUT/${TAG}_bench_c3b_60k.json 2> $OUT/bench_c3b.err || { echo "bench c3b failed"; exit 1; }
timeout -k 10 300 python3 bench.py --model c5 --no-cpu-baseline > $OUT/${TAG}_bench_c5_131k.json 2> $OUT/bench_c5.err || { echo "bench c5 failed"; exit 1; }
timeout -k 10 300 python3 bench.py --model c5 --dtype float64 --no-cpu-baseline > $OUT/${TAG}_bench_c5_131k_f64.json 2> $OUT/bench_c5_f64.err || { echo "bench c5 f64 failed"; exit 1; }
# 2. rocprofv3 kernel trace + stats of the same default command
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_bench_c3a_60k.csv
# 3. HBM traffic counters (two passes) and the MFMA utilisation pass on the deep tree
cd $R
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1 || { echo "pmc traffic failed"; tail -5 $OUT/pmc_traffic.log; exit 1; }
cp $R/gpurun_out/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
bash tools/pmc_valu.sh > $OUT/pmc_valu.log 2>&1 || { echo "pmc valu failed"; tail -5 $OUT/pmc_valu.log; exit 1; }
cp $R/gpurun_out/pmc_valu.json $OUT/${TAG}_pmc_valu.json
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1 || { echo "pmc mfma failed"; tail -5 $OUT/pmc_mfma.log; exit 1; }
cp $R/gpurun_out/pmc_mfma.json $OUT/${TAG}_pmc_mfma_c5.json
rm -rf $OUT/stats
ls -la $OUT
