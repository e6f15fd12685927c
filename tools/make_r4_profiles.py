"""Copies the summaries of tools/r4_measure.sh (gpurun_out/r4m_*) into the tracked profiles/r04_* files (bench.py reads
`roofline.traffic` of the dense and BM25 blocks from them). FETCH_SIZE / WRITE_SIZE are KiB per launch; FETCH_SIZE is doubled for
the wide (16 B per lane) streaming reads per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); Infinity-Cache hits are
counted in it. Parts whose inputs are missing are skipped."""
import csv
import json
import os
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def have(*names):
    return all(os.path.exists(os.path.join(O, n)) for n in names)


def load(name):
    return json.load(open(os.path.join(O, name)))


def stats(name):
    return {r["Name"]: r for r in csv.DictReader(open(os.path.join(O, name)))}


def short(k):
    for s in ("mx_gemm_kernel<mx_epi_qkv", "mx_gemm_kernel<mx_epi_ln", "mx_gemm_kernel<mx_epi_gelu", "ce_attention_kernelILi2ELb1ELb0", "ce_attention_kernelILi1ELb1ELb0", "ce_attention_kernelILi1ELb1ELb1",
              "mx_embed_ln_kernel", "mx_pool_classify_kernel", "mx_gather_rows_kernel", "ce_pack", "ce_pad"):
        if s in k:
            return {"ce_attention_kernelILi2ELb1ELb0": "ce_attention_kernel<2, MX>", "ce_attention_kernelILi1ELb1ELb0": "ce_attention_kernel<1, MX>", "ce_attention_kernelILi1ELb1ELb1": "ce_attention_kernel<1, MX, DIRECT> ([CLS] block of the last layer)"}.get(s, s + (">" if "<" in s else ""))
    return k[:60]


# ---- bench line
if have("r4m_bench.json"):
    line = [ln for ln in open(os.path.join(O, "r4m_bench.json")) if ln.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(P, "r04_bench_line.json"), "w"), indent=1)

if have("r4m_full_line_kernel_stats.csv"):
    shutil.copy(os.path.join(O, "r4m_full_line_kernel_stats.csv"), os.path.join(P, "r04_full_line_kernel_stats.csv"))
    if have("r4m_full_line_bench.json") and os.path.getsize(os.path.join(O, "r4m_full_line_bench.json")) > 2:
        json.dump(json.loads(open(os.path.join(O, "r4m_full_line_bench.json")).read()), open(os.path.join(P, "r04_full_line_bench_under_rocprof.json"), "w"), indent=1)

# ---- cross-encoder: kernel stats, traffic, SQ counters of `bench.py --mode rerank`
if have("r4m_ce_kernel_stats.csv"):
    shutil.copy(os.path.join(O, "r4m_ce_kernel_stats.csv"), os.path.join(P, "r04_ce_kernel_stats.csv"))
    if have("r4m_ce_timeline.txt"):
        shutil.copy(os.path.join(O, "r4m_ce_timeline.txt"), os.path.join(P, "r04_ce_forward_timeline.txt"))
if have("r4m_ce_pmc_FETCH_SIZE.json", "r4m_ce_pmc_WRITE_SIZE.json"):
    f, w = load("r4m_ce_pmc_FETCH_SIZE.json"), load("r4m_ce_pmc_WRITE_SIZE.json")
    emb = next(v for k, v in f.items() if "mx_embed_ln_kernel" in k)["FETCH_SIZE"]["launches"]
    forwards = emb / 4.0                                          # 25,600 pairs = 4 activation chunks, one embedding launch each
    kern, rd, wr = {}, 0.0, 0.0
    for k, v in f.items():
        if not any(s in k for s in ("mx_", "ce_")):
            continue
        fe, we = v["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", {"mean": 0.0, "launches": 0})
        r_b, w_b = fe["mean"] * 1024 * 2 * fe["launches"] / forwards, we["mean"] * 1024 * we["launches"] / forwards
        kern[short(k)] = {"launches_per_forward": fe["launches"] / forwards, "read_bytes_per_forward_x2": r_b, "write_bytes_per_forward": w_b,
                          "read_bytes_per_launch_x2": fe["mean"] * 1024 * 2, "write_bytes_per_launch": we["mean"] * 1024}
        rd += r_b
        wr += w_b
    json.dump({"source": "tools/r4_measure.sh cepmc on one MI355X (round 4): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of "
                         "`python3 bench.py --mode rerank --steps 20` (%g forwards of 25,600 pairs = 4 activation chunks each, 6 layers, the MX forward with the "
                         "[CLS]-only last layer). KiB per launch; FETCH_SIZE x 2 per MI355X_MICROARCH.md; Infinity-Cache hits are counted in FETCH_SIZE." % forwards,
               "per_forward_bytes": {"read_corrected_x2": rd, "write": wr, "total": rd + wr},
               "algorithmic_note": "stored activations are 3 B per element (hi16 + lo8; Q, K, V 4 B as fp16 pairs for the attention kernel): per token and "
                                   "full layer QKV 1.15 in + 4.6 out, attention 4.6 in + 1.15 out, out-proj 1.15 + 1.15 (residual) in + 1.15 out, FFN-up 1.15 "
                                   "in + 4.6 out, FFN-down 4.6 + 1.15 in + 1.15 out = 27.7 KB (round 3, fused FFN: 33 KB at 4 B per element)",
               "kernels": kern}, open(os.path.join(P, "r04_ce_traffic.json"), "w"), indent=1)
sq_files = [n for n in os.listdir(O) if n.startswith("r4m_cesq_pmc_") and n.endswith(".json")] if os.path.isdir(O) else []
if sq_files and have("r4m_ce_kernel_stats.csv"):
    ks = stats("r4m_ce_kernel_stats.csv")
    per = {}
    for n in sq_files:
        for k, v in load(n).items():
            per.setdefault(k, {}).update({cn: c["mean"] for cn, c in v.items()})
    out = {}
    for k, c in per.items():
        if "GRBM_GUI_ACTIVE" not in c or "mx_gemm" not in k and "attention" not in k:
            continue
        row = next((r for kk, r in ks.items() if kk[:60] == k[:60] or k[:40] in kk), None)
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        d = {"kernel_cycles_per_launch": cyc}
        if row:
            d["avg_us_per_launch_in_the_stats_pass"] = float(row["AverageUs"])
            d["clock_GHz_GRBM_over_duration"] = cyc / float(row["AverageUs"]) / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_pipe_busy_share"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
        if "SQ_WAVE_CYCLES" in c:
            d["wave_parked_share (SQ_WAIT_ANY)"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            d["issue_stalled_share (SQ_WAIT_INST_ANY)"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            d["issuing_share (SQ_ACTIVE_INST_ANY)"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_per_simd"] = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc
        if "SQ_LDS_IDX_ACTIVE" in c:
            d["lds_active_share_of_cu_cycles"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        d["raw"] = c
        out[short(k)] = d
    json.dump({"source": "tools/r4_measure.sh cesq: three rocprofv3 --pmc passes (SQ / GRBM counters, --kernel-trace only) of `python3 bench.py --mode rerank --steps 20`; "
                         "means per launch over all launches of a kernel (the small [CLS]-tail launches of the last layer pull the means of mx_gemm_kernel<ln | gelu> "
                         "down: read the qkv row for the main loop). Profiled passes run at a lower clock than un-profiled ones (guide, DVFS item 2).",
               "kernels": out}, open(os.path.join(P, "r04_ce_sq_counters.json"), "w"), indent=1)

# ---- dense: the structure bench.py reads (kernels -> dense_emit_kernel<false> -> hbm_traffic_bytes_per_launch -> total)
if have("r4m_dense_pmc_FETCH_SIZE.json", "r4m_dense_pmc_WRITE_SIZE.json"):
    f, w = load("r4m_dense_pmc_FETCH_SIZE.json"), load("r4m_dense_pmc_WRITE_SIZE.json")
    t = load("r4m_dense_pmc_TCC_HIT_sum_TCC_MISS_sum.json") if have("r4m_dense_pmc_TCC_HIT_sum_TCC_MISS_sum.json") else {}
    kern = {}
    for k, v in f.items():
        name = "dense_emit_kernel<false>" if "ILb0ELb0ELb0E" in k else ("dense_emit_kernel<true> (stage 0)" if "ILb1E" in k else k[:50])
        we = w.get(k, {}).get("WRITE_SIZE", {"mean": 0.0})
        d = {"FETCH_SIZE": v["FETCH_SIZE"], "WRITE_SIZE": we,
             "hbm_traffic_bytes_per_launch": {"read_corrected_x2": v["FETCH_SIZE"]["mean"] * 1024 * 2, "write": we["mean"] * 1024,
                                              "total": v["FETCH_SIZE"]["mean"] * 1024 * 2 + we["mean"] * 1024}}
        if k in t:
            d["l2_hit_rate"] = t[k]["TCC_HIT_sum"]["mean"] / (t[k]["TCC_HIT_sum"]["mean"] + t[k]["TCC_MISS_sum"]["mean"])
        kern[name] = d
    json.dump({"source": "tools/r4_measure.sh densepmc on one MI355X (round 4, the build whose bench line is profiles/r04_bench_line.json): rocprofv3 --pmc FETCH_SIZE | "
                         "WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, each its own pass with --kernel-trace only, of `python3 bench.py --mode dense --steps 20 "
                         "--no-cpu-baseline` (1M x 1536, 1,024-query batches); KiB per launch; FETCH_SIZE x 2 per MI355X_MICROARCH.md; means over the three "
                         "threshold-stage launches of a step (14k / 115k / 869k rows)", "kernels": kern}, open(os.path.join(P, "r04_dense_pmc.json"), "w"), indent=1)
    if have("r4m_dense_kernel_stats.csv"):
        shutil.copy(os.path.join(O, "r4m_dense_kernel_stats.csv"), os.path.join(P, "r04_dense_kernel_stats.csv"))

# ---- BM25: `python3 bench.py --mode hybrid --only-hybrid-calls --steps 6 --warmup 1`: 7 rag_hybrid_rrf_dev calls of 1024 queries
if have("r4m_hybrid_kernel_stats.csv", "r4m_hybrid_pmc_FETCH_SIZE.json", "r4m_hybrid_pmc_WRITE_SIZE.json", "r4m_hybrid_stats.log"):
    shutil.copy(os.path.join(O, "r4m_hybrid_kernel_stats.csv"), os.path.join(P, "r04_hybrid_kernel_stats.csv"))
    line = json.loads([ln for ln in open(os.path.join(O, "r4m_hybrid_stats.log")) if ln.startswith("{")][-1])
    calls = line["steps"] + line["warmup"]
    ks = stats("r4m_hybrid_kernel_stats.csv")
    fetch, write = load("r4m_hybrid_pmc_FETCH_SIZE.json"), load("r4m_hybrid_pmc_WRITE_SIZE.json")
    tcc = load("r4m_hybrid_pmc_TCC_HIT_sum_TCC_MISS_sum.json") if have("r4m_hybrid_pmc_TCC_HIT_sum_TCC_MISS_sum.json") else {}
    pick = lambda d, sub: next((v for k, v in d.items() if sub in k), None)
    kern = {}
    for s in ("bm25_range_kernel", "bm25_merge_select_kernel", "bm25_plan_kernel"):
        st, f_, w_, t_ = pick(ks, s), pick(fetch, s), pick(write, s), pick(tcc, s)
        if not (st and f_ and w_):
            continue
        per_call = f_["FETCH_SIZE"]["launches"] / calls
        kern[s] = {"launches_per_call": per_call, "avg_us_per_launch": float(st["AverageUs"]), "us_per_call": float(st["TotalDurationUs"]) / calls,
                   "fetch_bytes_per_call_x2": f_["FETCH_SIZE"]["mean"] * 1024 * 2 * per_call, "write_bytes_per_call": w_["WRITE_SIZE"]["mean"] * 1024 * per_call}
        if t_:
            kern[s]["l2_hit_rate"] = t_["TCC_HIT_sum"]["mean"] / (t_["TCC_HIT_sum"]["mean"] + t_["TCC_MISS_sum"]["mean"])
    sq = {}
    for n in os.listdir(O):
        if n.startswith("r4m_hybridsq_pmc_") and n.endswith(".json"):
            for name, v in load(n).items():
                if "bm25_range_kernel" in name:
                    sq.update({cn: c["mean"] for cn, c in v.items()})
    if "GRBM_GUI_ACTIVE" in sq:
        cyc = sq["GRBM_GUI_ACTIVE"] / 8.0
        sq["kernel_cycles"] = cyc
        for key, cn in (("valu_busy_per_simd", "SQ_ACTIVE_INST_VALU"), ("salu_busy_per_simd", "SQ_ACTIVE_INST_SCA"), ("lds_busy_per_simd", "SQ_ACTIVE_INST_LDS")):
            if cn in sq:
                sq[key] = sq[cn] * 4 / 1024 / cyc
        if "SQ_WAVE_CYCLES" in sq:
            sq["wait_any_share"] = sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
    us = sum(k["us_per_call"] for k in kern.values())
    traffic = sum(k["fetch_bytes_per_call_x2"] + k["write_bytes_per_call"] for k in kern.values())
    alg = line["bm25"]["postings_touched_per_batch"] * 12.0
    json.dump({"source": "tools/r4_measure.sh bm25pmc on one MI355X (round 4): rocprofv3 --kernel-trace --stats and separate --pmc passes of `python3 bench.py --mode "
                         "hybrid --only-hybrid-calls --steps 6 --warmup 1` (RAG_NO_FORK=1: the BM25 leg in line): every BM25 launch belongs to one of 7 "
                         "rag_hybrid_rrf_dev calls of 1024 queries (1M docs, nnz 9.5e7). FETCH_SIZE x 2 per MI355X_MICROARCH.md.",
               "bench_line_of_the_profiled_run": {"value": line["value"], "ms_per_step": line["ms_per_step"]},
               "kernels": kern,
               "per_call": {"bm25_device_us_under_rocprof": us, "algorithmic_bytes": alg, "hbm_traffic_bytes": traffic,
                            "traffic_over_algorithmic": traffic / alg, "algorithmic_GBs": alg / us / 1e3, "traffic_GBs": traffic / us / 1e3,
                            "l2_roof_frac (algorithmic bytes / 34.5 TB/s)": alg / us / 1e3 / 34500.0, "hbm_roof_frac (counter bytes / 8 TB/s)": traffic / us / 1e3 / 8000.0},
               "sq_counters_range_kernel": sq}, open(os.path.join(P, "r04_bm25_pmc.json"), "w"), indent=1)
print("profiles written:", sorted(n for n in os.listdir(P) if n.startswith("r04_")))
