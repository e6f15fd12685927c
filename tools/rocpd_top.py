"""Print the per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd sqlite database as CSV."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f'"{name}",{calls},{total:.3f},{avg:.3f},{pct:.4f}')
