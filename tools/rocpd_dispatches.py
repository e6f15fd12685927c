"""Per-dispatch durations (us), in launch order, of the kernels whose name contains <substring>, from a rocprofv3 rocpd sqlite
database (--kernel-trace). Usage: python tools/rocpd_dispatches.py <results.db> <substring> [max_rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 64
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print("tables/views:", tabs)
    sys.exit(1)
cols = [d[0] for d in db.execute(f"select * from {view} limit 1").description]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, start, end, grid_x, grid_y from {view} where {name_col} like ? order by start", (f"%{sub}%",)).fetchall()
for i, (n, s, e, gx, gy) in enumerate(rows[:limit]):
    print(f"{i:4d} {(e - s) / 1e3:10.2f} us  grid {gx} x {gy}  {n[:60]}")
