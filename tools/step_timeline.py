"""Print the kernel timeline of the last benchmark step from a rocprofv3 kernel-trace CSV:
start offset, duration and the idle gap before each kernel."""
import csv, glob, sys

path = sorted(glob.glob(sys.argv[1]))[-1]
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last evidence_scatter on
idx = max(i for i, r in enumerate(rows) if "evidence_scatter" in r["Kernel_Name"])
prev_idx = max(i for i, r in enumerate(rows[:idx]) if "evidence_scatter" in r["Kernel_Name"])
step = rows[prev_idx:idx]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:70]}")
    busy += e - s
    prev_end = e
print(f"step span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, kernels {len(step)}")
