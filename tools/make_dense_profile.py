"""Build the committed dense-kernel profile artefacts from rocprofv3 (rocpd sqlite) outputs of `python3 bench.py ...`:
  python tools/make_dense_profile.py <stats.db> <pmc_fetch.db> <pmc_l2.db> <pmc_sq.db> <bench.log>
-> profiles/r01_b_dense_kernel_stats.csv, profiles/r01_c_dense_pmc.json, profiles/r01_bench_line.json"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats_db, fetch_db, l2_db, sq_db, bench_log = sys.argv[1:6]
KERNEL = "dense_emit_kernelILb0ELb0EE"          # thresholded GEMM, full-batch variant


def counters(path):
    db = sqlite3.connect(path)
    acc = defaultdict(lambda: defaultdict(list))
    for name, disp, cn, v in db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                                        "group by kernel_name, dispatch_id, counter_name"):
        acc[name][cn].append(v)
    return acc


db = sqlite3.connect(stats_db)
with open(os.path.join(ROOT, "profiles", "r01_b_dense_kernel_stats.csv"), "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        f.write(f'"{name}",{calls},{total * 1000:.0f},{avg * 1000:.3f},{pct:.4f}\n')

merged = defaultdict(dict)
for p in (fetch_db, l2_db, sq_db):
    for name, d in counters(p).items():
        for cn, v in d.items():
            merged[name][cn] = {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)}
out = {"source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | SQ_* GRBM_GUI_ACTIVE) on `python3 bench.py "
                 "--steps 3 --warmup 1 --no-cpu-baseline` (MI355X, round 1, final kernels); FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads)", "kernels": {}}
for name, d in merged.items():
    key = "dense_emit_kernel<false>" if KERNEL in name else name[:100]
    out["kernels"][key] = dict(d)
k = out["kernels"].get("dense_emit_kernel<false>")
if k:
    rd = k["FETCH_SIZE"]["mean"] * 1024 * 2
    wr = k["WRITE_SIZE"]["mean"] * 1024
    k["hbm_traffic_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
    k["l2_hit_rate"] = k["TCC_HIT_sum"]["mean"] / (k["TCC_HIT_sum"]["mean"] + k["TCC_MISS_sum"]["mean"])
    gui = k["GRBM_GUI_ACTIVE"]["max"] / 8.0
    mf = k["SQ_VALU_MFMA_BUSY_CYCLES"]["max"] / 1024.0
    k["largest_launch"] = {"gui_active_cycles_per_xcd": gui, "mfma_busy_cycles_per_simd": mf, "mfma_pipe_util": mf / gui}
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_c_dense_pmc.json"), "w"), indent=1)
line = [l for l in open(bench_log) if l.startswith("{")][-1]
open(os.path.join(ROOT, "profiles", "r01_bench_line.json"), "w").write(line)
print(json.dumps({kk: vv for kk, vv in (k or {}).items() if kk in ("hbm_traffic_bytes_per_launch", "l2_hit_rate", "largest_launch")}, indent=1))
