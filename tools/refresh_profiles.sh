#!/bin/bash
# Everything profiles/ holds for one round, from the current tree, on the GPU box:
#   BN_GIT_HEAD=<commit> bash tools/refresh_profiles.sh r3
# Leaves the files under gpurun_out/profiles_<tag>/; copy them into profiles/ afterwards.
set -u
TAG=${1:-r1}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
# 1. the default bench line (with roofline, cpu_baseline and the two secondary workloads), plus the other models alone
timeout -k 10 400 python3 bench.py > $OUT/${TAG}_bench_c3a_60k.json 2> $OUT/bench_c3a.err || { echo "bench failed"; tail -5 $OUT/bench_c3a.err; exit 1; }
timeout -k 10 300 python3 bench.py --model c3b --no-cpu-baseline > $OUT/${TAG}_bench_c3b_60k.json 2> $OUT/bench_c3b.err || { echo "bench c3b failed"; exit 1; }
timeout -k 10 300 python3 bench.py --model c5 --no-cpu-baseline > $OUT/${TAG}_bench_c5_131k.json 2> $OUT/bench_c5.err || { echo "bench c5 failed"; exit 1; }
timeout -k 10 300 python3 bench.py --model c5 --dtype float64 --no-cpu-baseline > $OUT/${TAG}_bench_c5_131k_f64.json 2> $OUT/bench_c5_f64.err || { echo "bench c5 f64 failed"; exit 1; }
timeout -k 10 300 python3 bench.py --evidence resident-uint8 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_c3a_60k_uint8.json 2> $OUT/bench_u8.err || { echo "bench uint8 failed"; exit 1; }
# 2. rocprofv3 kernel trace + stats of the same default command
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-secondary > $OUT/stats.log 2>&1 || { echo "rocprof stats failed"; tail -5 $OUT/stats.log; exit 1; }
cp $OUT/stats/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats_bench_c3a_60k.csv
rm -rf $OUT/stats
# 3. counter passes of the three models' default commands: HBM traffic (two passes each), issue-side
#    counters, and the MFMA utilisation of the deep tree; each summary stamped with the commit and the
#    kernels' signatures of this build
cd $R
for M in c3a c3a_u8 c3b c5; do
  if [ $M = c3a ]; then SUF=""; ARGS=""; elif [ $M = c3a_u8 ]; then SUF="_c3a_u8"; ARGS="--evidence resident-uint8"; else SUF="_$M"; ARGS="--model $M"; fi
  PMC_TAG=$SUF bash tools/pmc_traffic.sh $ARGS > $OUT/pmc_traffic$SUF.log 2>&1 || { echo "pmc traffic $M failed"; tail -5 $OUT/pmc_traffic$SUF.log; exit 1; }
  cp $R/gpurun_out/pmc_traffic$SUF.json $OUT/${TAG}_pmc_traffic$SUF.json
  PMC_TAG=$SUF bash tools/pmc_valu.sh $ARGS > $OUT/pmc_valu$SUF.log 2>&1 || { echo "pmc valu $M failed"; tail -5 $OUT/pmc_valu$SUF.log; exit 1; }
  cp $R/gpurun_out/pmc_valu$SUF.json $OUT/${TAG}_pmc_valu$SUF.json
done
bash tools/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1 || { echo "pmc mfma failed"; tail -5 $OUT/pmc_mfma.log; exit 1; }
cp $R/gpurun_out/pmc_mfma.json $OUT/${TAG}_pmc_mfma_c5.json
# (the passes' raw counter files are large: only the summaries are kept)
rm -rf $R/gpurun_out/pmc $R/gpurun_out/pmc_* 2>/dev/null
python3 tools/pmc_meta.py $OUT/${TAG}_bench_c3a_60k.json $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_pmc_valu.json $OUT/${TAG}_pmc_traffic_c3b.json $OUT/${TAG}_pmc_valu_c3b.json $OUT/${TAG}_pmc_traffic_c5.json $OUT/${TAG}_pmc_valu_c5.json $OUT/${TAG}_pmc_mfma_c5.json
# (the same profile labels stand for other kernels on resident state bytes: stamped from that command's own line)
python3 tools/pmc_meta.py $OUT/${TAG}_bench_c3a_60k_uint8.json $OUT/${TAG}_pmc_traffic_c3a_u8.json $OUT/${TAG}_pmc_valu_c3a_u8.json
# 4. the EM step at other numbers of latent classes (cardinality packs) and the batch-size matrix
timeout -k 10 500 python3 tools/latent_classes.py 5 8 10 11 12 13 16 20 24 32 > $OUT/${TAG}_latent_classes.json 2> $OUT/latent.err || { echo "latent classes failed"; tail -3 $OUT/latent.err; exit 1; }
timeout -k 10 600 python3 tools/bench_matrix.py > $OUT/${TAG}_bench_matrix.json 2> $OUT/matrix.err || { echo "bench matrix failed"; tail -3 $OUT/matrix.err; exit 1; }
ls -la $OUT
