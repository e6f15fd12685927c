"""Launch-order timeline of the LAST n dispatches of a rocprofv3 rocpd database: start offset, duration and the idle gap before
each kernel (us). Usage: python tools/rocpd_timeline.py <results.db> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
t0, prev = rows[0][1], None
for nm, s, e in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f}  gap {gap:8.1f}  {nm[:70]}")
    prev = e
