"""The per-GPU share of BASELINE.json configs[4] ("100M x 1536-d corpus sharded 8-way, hybrid + rerank, per-shard top-k
RCCL all-gather over xGMI"): 12.5M rows per GPU with their fp32 masters, fp16 MFMA operand, doc-partitioned BM25 postings
over a multi-million-term vocabulary, the replicated passage token store and the cross-encoder - built on ONE rank
(world = 1: the single-GPU share, tools/r3_shard.sh and tests/test_full_size_gpu.py) or on every rank of a node
(bench.py --gpus N). Everything synthetic is generated on the GPU with torch (data generation only, never the measured
path) in seeded chunks, so the oracle side of a parity check can regenerate any piece.

Reference path this stands for: /root/reference/rag/document_store.py:448-460 (pgvector top-k), rag/retrieval.py:324-347
(BM25Okapi over the corpus), rag/reranker.py:224-271 (RRF) and :346-359 (cross-encoder rerank).
"""
import time

import numpy as np
import torch

import bench as BE
import bench_modes as BM

DIM = BE.DIM
VOCAB = 2_000_000          # distinct term ids of the synthetic text (truncated Zipf(1.1)): the tail `doc.lower().split()` has
MEAN_LEN = 120
TOK_L = 224                # passage tokens kept per document (~U[96, 224])
TOK_CHUNK = 500_000


def gen_tokens_chunk(c, rows, device, vocab_size, L=TOK_L):
    """Rows [c*TOK_CHUNK, ...) of the passage token store: WordPiece ids ~U[1000, vocab), lengths ~U[96, L]."""
    g = torch.Generator(device=device)
    g.manual_seed(5000 + c)
    tok = torch.randint(1000, vocab_size, (rows, L), generator=g, device=device, dtype=torch.int32)
    ln = torch.randint(96, L + 1, (rows,), generator=g, device=device, dtype=torch.int32)
    return tok, ln


def build_shard(eng, device, rows_per_gpu, rank=0, world=1, Q=256, vocab=VOCAB, ce_seed=2024, with_rerank=True, log=None):
    """Loads this rank's share into `eng` and returns the replicated query batch + what an oracle check needs.
    Global doc ids: rank r owns [r * rows_per_gpu, (r + 1) * rows_per_gpu). BM25 statistics (df -> idf, avgdl) are GLOBAL:
    summed over the ranks with one all-reduce each, as SURVEY 8e prescribes (computed at index build, replicated)."""
    import torch.distributed as dist
    from optimized_rag_amd.bm25 import Bm25Postings
    from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors
    say = log or (lambda *a: None)
    total_rows = rows_per_gpu * world
    chunk = BE.CHUNK_ROWS
    assert rows_per_gpu % chunk == 0
    n_chunks = total_rows // chunk
    my_chunks = range(rank * (rows_per_gpu // chunk), (rank + 1) * (rows_per_gpu // chunk))
    t0 = time.perf_counter()
    eng.index_reserve(rows_per_gpu, id_base=rank * rows_per_gpu)
    for c in my_chunks:
        eng.index_append(BE.gen_chunk(c, chunk, device, "iid", total_rows))
    queries, planted = BE.gen_queries(Q, total_rows, n_chunks, chunk, device, "iid")
    say(f"dense index: {rows_per_gpu} rows in {time.perf_counter() - t0:.1f}s")
    # ---- postings of this rank's documents (local doc numbers), global statistics -----------------------------------
    t0 = time.perf_counter()
    qdocs = np.random.default_rng(7).integers(0, rows_per_gpu, Q)              # query terms are drawn from rank 0's documents
    indptr, d, tf, dl, sampled = BM.zipf_postings_gpu(rows_per_gpu, vocab, MEAN_LEN, device, seed=99 + rank, sample_docs=qdocs if rank == 0 else None)
    df = torch.from_numpy(np.diff(indptr)).to(device)
    len_sum = torch.tensor([float(dl.sum())], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(df)
        dist.all_reduce(len_sum)
    df_h = df.cpu().numpy()
    idf = Bm25Postings.idf_table(df_h, total_rows)
    idf[df_h == 0] = 0.0
    avgdl = float(len_sum.item()) / total_rows
    post = Bm25Postings(indptr, d, tf, dl, idf, avgdl)
    post.load(eng)
    if world > 1:
        eng.bm25_set_normalize(False)                                           # shards hand out RAW scores (sharded.py)
    if rank == 0:
        ptr, terms = BM.term_queries_from_docs(sampled, qdocs)
    if world > 1:
        box = [(ptr, terms) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ptr, terms = box[0]
    say(f"postings: nnz {int(indptr[-1])}, {int((np.diff(indptr) > 0).sum())} distinct terms locally, in {time.perf_counter() - t0:.1f}s")
    out = dict(queries=queries, planted=planted, post=post, term_ptr=ptr, terms=terms,
               ptr_d=torch.from_numpy(ptr).to(device), terms_d=torch.from_numpy(terms).to(device), total_rows=total_rows)
    if with_rerank:
        t0 = time.perf_counter()
        cfg = MINILM_L6_CONFIG
        tensors = random_init_tensors(cfg, ce_seed)
        eng.ce_load(cfg, tensors)
        # REPLICATED passage token store over all `total_rows` documents (100M x 224 x 2 B = 44.8 GB fits)
        eng.tokens_reserve(total_rows, TOK_L)
        for c in range((total_rows + TOK_CHUNK - 1) // TOK_CHUNK):
            tok, ln = gen_tokens_chunk(c, min(TOK_CHUNK, total_rows - c * TOK_CHUNK), device, cfg["vocab_size"])
            eng.tokens_append_dev(tok, ln)
        g = torch.Generator()
        g.manual_seed(8)
        q_tok = torch.randint(1000, cfg["vocab_size"], (Q, 16), generator=g, dtype=torch.int32)
        out.update(cfg=cfg, tensors=tensors, q_tok=q_tok, q_tok_d=q_tok.to(device),
                   q_len_d=torch.full((Q,), 16, dtype=torch.int32, device=device))
        say(f"cross-encoder + {total_rows}-passage token store in {time.perf_counter() - t0:.1f}s")
    torch.cuda.empty_cache()
    return out


def hbm_used_gb(device):
    free, total = torch.cuda.mem_get_info(device)
    return round((total - free) / 1e9, 1), round(total / 1e9, 1)


def shard_blocks(eng, st, device, rank=0, world=1, steps=3, pool=100, k=20, L_pair=256, with_rerank=True):
    """Times the sharded classes on the loaded share: dense (1024 / 256 / 128 queries), hybrid (256), retrieve + rerank (256).
    bench.py's contract per timing: barrier + synchronize on both sides, max over ranks."""
    from optimized_rag_amd.sharded import ShardedDenseIndex, ShardedHybridIndex, ShardedPipeline
    q = st["queries"]
    Q = q.shape[0]
    blocks = {}
    dense = ShardedDenseIndex(eng, rank=rank, world=world)
    for nq in sorted({Q, min(Q, 128)}, reverse=True):
        qq = q[:nq].contiguous()
        t = BM.timed_all_ranks(lambda: dense.search(qq, k), steps, 1, world)
        ids, _ = dense.search(qq, k)
        torch.cuda.synchronize()
        hit = float((ids[:, 0].cpu() == st["planted"][:nq]).float().mean())
        n_local = st["total_rows"] // world
        blocks[f"dense_q{nq}"] = {"queries_per_sec": round(nq / t, 1), "ms_per_batch": round(t * 1e3, 3),
                                  "planted_neighbour_at_rank1": hit,
                                  "corpus_pass_GBs_per_gpu": round(n_local * DIM * 2 / t / 1e9, 1),
                                  "hbm_roof_frac": round(n_local * DIM * 2 / t / 1e9 / BE.PEAK_HBM_GBS, 4),
                                  "mfma_roof_frac": round(2.0 * nq * n_local * DIM / t / 1e12 / BE.PEAK_MFMA_TFLOPS, 4)}
    # one GPU: the one-call entries (rag_hybrid_rrf_dev / rag_retrieve_rerank_dev); several: the sharded classes around them
    hyb = ShardedHybridIndex(eng, rank=rank, world=world)
    hybrid = (lambda: eng.hybrid_rrf_dev(q, st["ptr_d"], st["terms_d"], pool, k)) if world == 1 else \
             (lambda: hyb.search_hybrid(q, st["ptr_d"], st["terms_d"], pool, k))
    t = BM.timed_all_ranks(hybrid, steps, 1, world)
    blocks["hybrid_q%d" % Q] = {"queries_per_sec": round(Q / t, 1), "ms_per_batch": round(t * 1e3, 3),
                                "workload": f"dense top-{pool} + BM25 top-{pool} + RRF -> top-{k}" +
                                            ("" if world == 1 else ", one all-gather of both lists")}
    # the BM25 leg of that call alone on this rank's share (device time from HIP events on the launch stream), priced like the 1M-row
    # `hybrid` block: algorithmic posting bytes against the L2 roof (the kernel is issue-bound, DESIGN 4.2: neither byte roof binds)
    try:
        ids_d = torch.empty((Q, pool), dtype=torch.int64, device=device)
        sc_d = torch.empty((Q, pool), dtype=torch.float64, device=device)
        eng.bm25_topk_dev(st["ptr_d"], st["terms_d"], pool, ids_d, None, sc_d)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        for _ in range(steps):
            eng.bm25_topk_dev(st["ptr_d"], st["terms_d"], pool, ids_d, None, sc_d)
        torch.cuda.synchronize()
        bm_ms, bm_spans = eng.stage_kernel_ms(1)
        eng.set_profiling(False)
        counts = np.diff(st["post"].indptr)
        tq = np.asarray(st["terms"])
        tq = tq[(tq >= 0) & (tq < counts.shape[0])]
        bm_bytes = float(counts[tq].sum()) * 12.0
        leg = bm_ms / max(1, bm_spans)
        blocks["hybrid_q%d" % Q]["bm25_leg"] = {
            "ms_per_batch_on_this_share": round(leg, 3), "algorithmic_bytes": bm_bytes,
            "l2_roof_frac": round(bm_bytes / (leg * 1e-3) / 1e9 / BM.PEAK_L2_GBS, 4),
            "bound": "issue (vector + scalar ALU of bm25_range_kernel; postings served by the L2s)"}
    except Exception as e:                                   # a secondary figure must not cost the line
        blocks["hybrid_q%d" % Q]["bm25_leg"] = {"error": repr(e)[:200]}
    if with_rerank:
        pipe = ShardedPipeline(eng, rank=rank, world=world)

        def run():
            if world == 1:
                return eng.retrieve_rerank_dev(q, st["q_tok_d"], st["q_len_d"], pool, k, term_ptr=st["ptr_d"], terms=st["terms_d"], L_pair=L_pair)
            return pipe.retrieve_rerank(q, st["ptr_d"], st["terms_d"], st["q_tok_d"], st["q_len_d"], pool, k, L_pair=L_pair)

        t = BM.timed_all_ranks(run, max(1, steps - 1), 1, world)
        blocks["retrieve_rerank_q%d" % Q] = {"queries_per_sec": round(Q / t, 2), "ms_per_batch": round(t * 1e3, 2),
                                             "workload": f"hybrid top-{pool} -> MiniLM-L-6 cross-encoder (L={L_pair}, pairs split over the "
                                                         f"ranks) -> top-{k}"}
        if world == 1:
            # algorithmic FLOPs of the scored pairs (SURVEY 8d per-pair formula on their real token counts) over the WHOLE batch time,
            # retrieval included: a lower bound of the forward's own fraction (the headline block prices the forward alone)
            try:
                res = run()
                torch.cuda.synchronize()
                plens, _ = token_lengths(res[3].cpu().numpy().reshape(-1), st["total_rows"], device, st["cfg"]["vocab_size"])
                plen = np.minimum(16 + plens.astype(np.float64) + 3, L_pair)
                flops = float((6.0 * plen * (3.539e6 + 1536.0 * plen)).sum())
                blocks["retrieve_rerank_q%d" % Q]["mfma_roof_frac_of_the_whole_batch"] = round(flops / t / 1e12 / BE.PEAK_MFMA_TFLOPS, 4)
                blocks["retrieve_rerank_q%d" % Q]["algorithmic_flops_per_batch"] = flops
            except Exception as e:
                blocks["retrieve_rerank_q%d" % Q]["mfma_roof_frac_of_the_whole_batch"] = None
                blocks["retrieve_rerank_q%d" % Q]["roofline_error"] = repr(e)[:200]
        # BASELINE.json's second metric on this configuration: p50 latency of ONE query through retrieve + rerank (every rank
        # searches its shard, the 100 pairs are split over the ranks), synchronised per call, max over the ranks per call
        q1, ptr1 = q[:1].contiguous(), st["ptr_d"][:2].contiguous()
        tok1, len1 = st["q_tok_d"][:1].contiguous(), st["q_len_d"][:1].contiguous()

        def one():
            if world == 1:
                return eng.retrieve_rerank_dev(q1, tok1, len1, pool, k, term_ptr=ptr1, terms=st["terms_d"], L_pair=L_pair)
            return pipe.retrieve_rerank(q1, ptr1, st["terms_d"], tok1, len1, pool, k, L_pair=L_pair)

        import torch.distributed as dist
        lat = []
        for i in range(3 + 21):                              # 3 warm-ups
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            one()
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)    # outside the timed region
            if i >= 3:
                lat.append(float(dt.item()))
        lat.sort()
        blocks["retrieve_rerank_single_query_p50_ms"] = round(lat[len(lat) // 2] * 1e3, 3)
    used, total = hbm_used_gb(device)
    blocks["hbm_used_gb"] = used
    blocks["hbm_total_gb"] = total
    return blocks


def token_lengths(rows, total_rows, device, vocab_size, with_tokens=False):
    """Lengths (and optionally the token rows) of the given documents of the replicated store, regenerated chunk by chunk with the
    store's own generator (the store itself lives inside the library)."""
    rows = np.asarray(rows, dtype=np.int64)
    lens = np.zeros(rows.shape, dtype=np.int64)
    toks = np.zeros(rows.shape + (TOK_L,), dtype=np.int32) if with_tokens else None
    for c in np.unique(rows // TOK_CHUNK):
        tok, ln = gen_tokens_chunk(int(c), min(TOK_CHUNK, total_rows - int(c) * TOK_CHUNK), device, vocab_size)
        sel = (rows // TOK_CHUNK) == c
        idx = torch.from_numpy(rows[sel] - c * TOK_CHUNK).to(device)
        lens[sel] = ln[idx].cpu().numpy()
        if with_tokens:
            toks[sel] = tok[idx].cpu().numpy()
        del tok, ln
    return lens, toks


def headline_rerank(eng, device, rows_total, rank=0, world=1, steps=20, warmup=3, cpu_baseline=True, pool=100, k=20, L_pair=256, phase=None):
    """BASELINE.json configs[3], the configuration its metric is quoted on: hybrid top-100 -> MiniLM-L-6 cross-encoder -> top-20,
    256-query batches over a `rows_total` x 1536-d corpus, row-sharded over `world` ranks (strong scaling: the same corpus and
    batch at every N; one rank: the one-call entry rag_retrieve_rerank_dev, several: ShardedPipeline - per-shard candidate lists,
    one all-gather, fusion, the 25,600 pairs split over the ranks, one gather of the logits). bench.py's timing contract:
    `warmup` untimed batches, exactly `steps` timed ones between barrier + synchronize, MAX over ranks. Returns the fields of the
    bench line (value, ms_per_step, latencies, roofline, cpu_baseline)."""
    import torch.distributed as dist
    from optimized_rag_amd.sharded import ShardedPipeline
    say = phase or (lambda s: None)
    Q, Lq = 256, 16
    say("headline: build_shard")
    st = build_shard(eng, device, rows_total // world, rank=rank, world=world, Q=Q)
    q = st["queries"]
    pipe = ShardedPipeline(eng, rank=rank, world=world)

    def run(nq=Q):
        if world == 1:
            return eng.retrieve_rerank_dev(q[:nq], st["q_tok_d"][:nq], st["q_len_d"][:nq], pool, k, term_ptr=st["ptr_d"][:nq + 1], terms=st["terms_d"], L_pair=L_pair)
        return pipe.retrieve_rerank(q[:nq].contiguous(), st["ptr_d"][:nq + 1].contiguous(), st["terms_d"], st["q_tok_d"][:nq].contiguous(),
                                    st["q_len_d"][:nq].contiguous(), pool, k, L_pair=L_pair)

    say("headline: warmup")
    for _ in range(max(1, warmup)):
        run()
    torch.cuda.synchronize()
    eng.set_profiling(True)
    say("headline: timed steps")
    t = BM.timed_all_ranks(run, steps, 0, world)
    ce_ms, ce_spans = eng.stage_kernel_ms(2)
    eng.set_profiling(False)
    say("headline: latencies")
    lat_b, lat_1 = [], []
    for which, n_it, warm, nq in ((lat_b, 5, 0, Q), (lat_1, 60, 6, 1)):
        for i in range(warm + n_it):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            run(nq)
            torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            if i >= warm:
                which.append(float(dt.item()) * 1e3)
    res = run()
    torch.cuda.synchronize()
    ids, sc, cand = res[0], res[1], res[3]
    ok = bool((ids >= 0).all().item() and (sc[:, :-1] >= sc[:, 1:]).all().item())
    cand_h = cand.cpu().numpy()
    cfg = st["cfg"]
    # algorithmic FLOPs of one batch: SURVEY 8d per-pair formula 6 * len * (3.539e6 + 1536 * len) on the REAL token counts of the
    # pairs that were scored ([CLS] q [SEP] passage [SEP], longest_first to L); padding is neither computed nor counted
    plens, _ = token_lengths(cand_h.reshape(-1), rows_total, device, cfg["vocab_size"])
    plen = np.minimum(Lq + plens.astype(np.float64) + 3, L_pair)
    flops = float((6.0 * plen * (3.539e6 + 1536.0 * plen)).sum())
    fwd_ms = ce_ms / max(1, ce_spans)
    ce_tf = flops / world / (fwd_ms * 1e-3) / 1e12            # this rank's forward scores 1 / world of the pairs
    out = {
        "value": round(Q / t, 2), "ms_per_step": round(t * 1e3, 3),
        "p50_batch_latency_ms": round(float(np.median(lat_b)), 3), "p50_single_query_latency_ms": round(float(np.median(lat_1)), 3),
        "pairs_per_sec": round(Q * pool / t, 1), "mean_pair_tokens": round(float(plen.mean()), 1),
        "roofline": {"bound": "mfma", "kernel": "cross-encoder forward of one batch on rank 0 (mx_gemm_kernel<qkv | ln | gelu> + ce_attention_kernel + "
                                                  "embedding / pooler), all chunks; dominant kernel mx_gemm_kernel<mx_epi_gelu> (FFN up-projection)",
                     "achieved": round(ce_tf, 2), "peak": BE.PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ce_tf / BE.PEAK_MFMA_TFLOPS, 4),
                     "traffic": BM._profile_number("r04_ce_traffic.json", ("per_forward_bytes", "total")) if (world == 1 and rows_total == 1_000_000) else None,
                     "traffic_source": "profiles/r04_ce_traffic.json: HBM bytes of one 25,600-pair forward (FETCH_SIZE x 2 + WRITE_SIZE, separate "
                                       "rocprofv3 --pmc passes of `bench.py --mode rerank`; Infinity-Cache hits are counted in FETCH_SIZE)",
                     "avg_forward_ms": round(fwd_ms, 3), "algorithmic_flops_per_forward": flops / world,
                     "share_of_batch_time": round(fwd_ms / (t * 1e3), 4),
                     "note": "algorithmic FLOPs / device time of the forward (HIP events on the launch stream). Operands are hi16 + lo8: per "
                             "algorithmic product the matrix pipe issues one fp16 MFMA + one block-scaled bf8 MFMA at twice the rate = 2 "
                             "fp16-equivalent units (rounds 1-3: 3), so the pipe sees twice this rate"},
        "sanity": {"all_slots_filled_and_sorted": ok},
        "hbm_used_gb": hbm_used_gb(device)[0],
    }
    if cpu_baseline and rank == 0:
        say("headline: cpu baseline")
        from oracle.cpu_baseline import bert_cpu_pairs
        ns = 64                                              # 2 batches of 32 pairs = the first 64 candidates of query 0
        rows = cand_h[0, :ns]
        tl, tk = token_lengths(rows, rows_total, device, cfg["vocab_size"], with_tokens=True)
        q_tok = st["q_tok"]
        pid = np.zeros((ns, L_pair), dtype=np.int64)
        ptt = np.zeros((ns, L_pair), dtype=np.int64)
        pl = np.zeros(ns, dtype=np.int64)
        for j in range(ns):
            dlen = int(min(tl[j], L_pair - 3 - Lq))
            row = [101] + q_tok[0].tolist() + [102] + tk[j, :dlen].tolist() + [102]
            pid[j, :len(row)] = row
            ptt[j, Lq + 2:len(row)] = 1
            pl[j] = len(row)
        cl, cdt, threads = bert_cpu_pairs(cfg, st["tensors"], pid, ptt, pl, batch=32)
        out["cpu_baseline"] = {"value": round(ns / cdt / pool, 4), "unit": "queries/sec", "cores": threads, "kind": "port",
                               "pairs_per_sec": round(ns / cdt, 2),
                               "sample": f"rerank stage only (the retrieval legs have their own baselines in the `dense` and `hybrid` blocks): {ns} of "
                                         f"the batch's {Q * pool} pairs (the first {ns} candidates of query 0), torch-CPU "
                                         f"BertForSequenceClassification fp32, batch 32 padded to the longest pair (what sentence-transformers' "
                                         f"CrossEncoder.predict does on CPU), {cdt:.2f}s; value = pairs/s / {pool} pairs per query"}
    return out
