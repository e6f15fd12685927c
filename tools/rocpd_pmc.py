"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd sqlite database (one --pmc pass): JSON to stdout.
Usage: python tools/rocpd_pmc.py <results.db> [kernel-name-substring]"""
import json
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print(json.dumps({"error": "no counters_collection view", "tables": tabs}))
    sys.exit(1)
cols = [d[0] for d in c.execute(f"select * from {view} limit 1").description]
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select {name_col}, dispatch_id, counter_name, sum(value) from {view} group by {name_col}, dispatch_id, counter_name").fetchall()
acc = defaultdict(lambda: defaultdict(list))
for name, disp, cn, v in rows:
    if flt in name:
        acc[name][cn].append(v)
out = {k[:80]: {cn: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for cn, v in d.items()} for k, d in acc.items()}
print(json.dumps(out, indent=1))
