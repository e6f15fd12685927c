"""Copies the summaries of tools/r2_measure.sh (gpurun_out/r2m_*) into the tracked profiles/r02_d..i files."""
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def last_json(name):
    return json.loads(open(os.path.join(O, name)).read().strip().splitlines()[-1])


def pmc(name):
    return json.load(open(os.path.join(O, name)))


line = last_json("r2m_bench.json")
json.dump(line, open(os.path.join(P, "r02_d_bench_line.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "r2m_dense_kernel_stats.csv"), os.path.join(P, "r02_e_dense_kernel_stats.csv"))
shutil.copy(os.path.join(O, "r2m_full_kernel_stats.csv"), os.path.join(P, "r02_f_full_line_kernel_stats.csv"))

# dense PMC: one JSON, counters merged per kernel; FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports 1/2 of wide reads)
kern = {}
for f in sorted(os.listdir(O)):
    if f.startswith("r2m_pmc_") and f.endswith(".txt"):
        for k, v in pmc(f).items():
            short = ("dense_emit_kernel<false>" if "ILb0ELb0ELb0" in k
                     else ("dense_emit_kernel<true> (stage 0)" if "ILb1ELb0ELb0" in k else k[:60]))
            kern.setdefault(short, {}).update(v)
main = kern["dense_emit_kernel<false>"]
fetch_b = main["FETCH_SIZE"]["mean"] * 1024 * 2
write_b = main["WRITE_SIZE"]["mean"] * 1024
l2_hit = main["TCC_HIT_sum"]["mean"] / (main["TCC_HIT_sum"]["mean"] + main["TCC_MISS_sum"]["mean"])
# bench.py reads kernels["dense_emit_kernel<false>"]["hbm_traffic_bytes_per_launch"]["total"] into roofline.traffic
main["hbm_traffic_bytes_per_launch"] = {"read_corrected_x2": fetch_b, "write": write_b, "total": fetch_b + write_b}
main["l2_hit_rate"] = l2_hit
main["lds_bank_conflict_frac"] = main["SQ_LDS_BANK_CONFLICT"]["mean"] / max(1.0, main["SQ_LDS_IDX_ACTIVE"]["mean"])
json.dump({
    "source": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES), each its own pass with --kernel-trace only, on `python3 bench.py --dense-only "
              "--no-cpu-baseline --steps 5 --warmup 1 --latency-batches 1` (MI355X, round 2, tools/r2_measure.sh); FETCH_SIZE/WRITE_SIZE "
              "are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); means are over the "
              "three threshold-stage launches of a step (14k / 115k / 869k rows)",
    "kernels": {k: v for k, v in kern.items() if k.startswith("dense_emit")}},
    open(os.path.join(P, "r02_g_dense_pmc.json"), "w"), indent=1)

keep = ("value", "ms_per_step", "p50_batch_latency_ms", "exactness", "roofline")
variants = {c: {k: last_json(f"r2m_bench_{c}.json").get(k) for k in keep} for c in ("clustered", "sorted", "tenant-contiguous")}
ab = {n: {k: last_json(f"r2m_bench_{n}.json").get(k) for k in ("value", "ms_per_step")}
      for n in ("linear_order", "no_second_pass", "dense_only")}
json.dump({"source": "tools/r2_measure.sh on one MI355X box: `python bench.py --dense-only --corpus X` (1M x 1536, 1024 queries, "
                     "top-20); A/B on the same box with RAG_DENSE_LINEAR_ORDER=1 (r1's table order) and RAG_NO_SECOND_PASS=1",
           "iid": {k: line.get(k) for k in keep}, "row_order_variants": variants, "same_box_ab": ab},
          open(os.path.join(P, "r02_h_row_order_variants.json"), "w"), indent=1)

ce = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in pmc(f"r2m_cepmc_{c}.txt").items():
        ce.setdefault(k, {}).update(v)
forwards = 3          # bench.py --mode rerank runs 3 forwards: 72 launches per GEMM kernel = 3 forwards x 4 chunk launches x 6 layers
rd = sum(v["FETCH_SIZE"]["mean"] * v["FETCH_SIZE"]["launches"] for v in ce.values() if "FETCH_SIZE" in v) * 1024 * 2 / forwards
wr = sum(v["WRITE_SIZE"]["mean"] * v["WRITE_SIZE"]["launches"] for v in ce.values() if "WRITE_SIZE" in v) * 1024 / forwards
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) on `python3 bench.py --mode "
                     "rerank` (3 forwards of 25,600 pairs = 4 activation chunks each, 6 layers); KiB per launch; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md",
           "per_forward_bytes": {"read_corrected_x2": rd, "write": wr, "total": rd + wr,
                                 "algorithmic": {"per_token_and_layer_bytes": 33792, "tokens": 4.6e6, "layers": 6,
                                                 "total": 33792 * 4.6e6 * 6}},
           "kernels": ce}, open(os.path.join(P, "r02_i_ce_traffic.json"), "w"), indent=1)

hy, rr = line["hybrid"], line["retrieve_rerank"]
print("value", line["value"], "ms", line["ms_per_step"], "p50", line["p50_batch_latency_ms"], "frac", line["roofline"]["frac"],
      "avg_launch_ms", line["roofline"]["avg_launch_ms"], "achieved", line["roofline"]["achieved"])
print("traffic", fetch_b + write_b, "l2 hit", l2_hit)
print("hybrid", hy["value"], hy["ms_per_batch"], hy["queries_per_sec_batch256"], hy["p50_single_query_latency_ms"],
      hy["linear_fusion_queries_per_sec_batch256"], hy["roofline"], hy["cpu_baseline"]["value"])
print("rerank", rr["value"], rr["ms_per_batch"], rr["p50_single_query_latency_ms"], rr["roofline"]["achieved"], rr["roofline"]["frac"],
      rr["cpu_baseline"]["value"])
print("agent", line["agent_latency"]["p50_ms"], line["agent_latency"]["cpu_baseline"])
print("cpu", line["cpu_baseline"]["value"], line["cpu_baseline"]["cores"])
print({c: (v["value"], v["exactness"]["exact_scan"], v["exactness"]["overflowed"]) for c, v in variants.items()}, ab)
print("ce traffic per forward: read", rd / 1e9, "GB write", wr / 1e9, "GB")
