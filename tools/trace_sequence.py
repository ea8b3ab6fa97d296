"""Prints the kernels of the last EM step of a rocprofv3 kernel trace (csv) in launch order with durations and gaps."""
import csv, glob, re, sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "bn::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last but one m_step_rows
ends = [i for i, r in enumerate(rows) if "m_step_rows" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
prev = None
for r in rows[lo:ends[-1] + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    print("%-58s %8.1f us  gap %6.1f  grid %s wg %s vgpr %s lds %s" % (name[:58], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0,
          r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
    prev = e
