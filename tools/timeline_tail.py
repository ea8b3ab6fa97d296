"""From a rocprofv3 kernel-trace CSV: the last `count` kernels with start offset, idle gap before and duration."""
import csv, glob, sys

path = sorted(glob.glob(sys.argv[1]))[-1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # kernels to leave out at the end
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) - skip - count:len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
    prev_end = e
