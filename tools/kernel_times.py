"""Average duration per kernel symbol of a `rocprofv3 --kernel-trace` run (its rocpd database):

    cd /tmp && rocprofv3 --kernel-trace -d out -o x -- python3 $REPO/bench.py --batch 7500 --no-cpu-baseline --no-secondary
    python3 tools/kernel_times.py out [minimum launches] [minimum microseconds]

One line per kernel of this library (namespace bn) with at least that many launches."""
import glob
import sqlite3
import sys

path = glob.glob(sys.argv[1] + "/*.db")[0]
min_launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
db = sqlite3.connect(path)
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
dispatches = [t for t in tables if "kernel_dispatch" in t][0]
symbols = [t for t in tables if "kernel_symbol" in t][0]
query = (f"select s.kernel_name, count(*), avg(d.end - d.start) / 1e3 from {dispatches} d "
         f"join {symbols} s on d.kernel_id = s.id group by 1 order by 3 desc")
for name, launches, micros in db.execute(query):
    if launches >= min_launches and micros >= min_us and "bn" in name:
        print(f"{micros:9.1f} us  x{launches:<6d} {name[:110]}")
