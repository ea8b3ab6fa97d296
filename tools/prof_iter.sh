export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_b -- python3 $R/bench.py --steps 3 --warmup 1 --batch 60000 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?"; python3 - <<PY
import csv,glob,json
f=sorted(glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_b/*/*kernel_trace.csv"))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=[r for r in rows if r["Kernel_Name"].startswith("void bn::") or r["Kernel_Name"].startswith("bn::")]
for r in sel[-16:]:
    print(r["Kernel_Name"][:44], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
print(open("$GRAFT_REPO_ROOT/gpurun_out/bench_prof.log").read()[-1500:-1100])
PY
