"""Secondary bench modes (not the driver's headline line): BASELINE.json configs[2] (hybrid: dense + BM25(CSR) + RRF)
and configs[3] (cross-encoder rerank of the hybrid top-100). Single GPU. Prints one JSON line each.

  python bench.py --mode hybrid  [--rows 1000000 --queries 1024]
  python bench.py --mode rerank  [--queries 256]
"""
import json
import time

import numpy as np
import torch

DIM = 1536


def synthetic_csr(n_docs, vocab, mean_len, seed=99):
    """doc length ~ Poisson(mean_len), tokens Zipf(1.1) over `vocab` ids -> term-major CSR (docs ascending)."""
    rng = np.random.default_rng(seed)
    lens = rng.poisson(mean_len, n_docs).astype(np.int64)
    total = int(lens.sum())
    tok = (rng.zipf(1.1, total) - 1) % vocab
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), lens)
    key = tok.astype(np.int64) * n_docs + doc
    uk, tf = np.unique(key, return_counts=True)
    term = (uk // n_docs).astype(np.int64)
    d = (uk % n_docs).astype(np.int32)
    indptr = np.zeros(vocab + 1, dtype=np.int64)
    np.add.at(indptr, term + 1, 1)
    indptr = np.cumsum(indptr)
    return indptr, d, tf.astype(np.int32), lens.astype(np.int32), tok, np.concatenate([[0], np.cumsum(lens)])


def zipf_postings_gpu(n_docs, vocab, mean_len, device, seed=99, s=1.1, chunk_docs=500_000, sample_docs=None):
    """Synthetic postings at shard scale, generated ON THE GPU (torch is used as a data generator here, never on the measured
    path): doc length ~ Poisson(mean_len), tokens drawn from a TRUNCATED Zipf(s) over `vocab` term ids (inverse CDF), so a
    multi-million-term vocabulary has the long tail of rare terms a real `doc.lower().split()` corpus has
    (rag/retrieval.py:334-335) - synthetic_csr's folded Zipf over 100k ids does not. Term-major CSR, docs ascending.
    Returns host arrays (indptr int64 [V+1], doc int32 [nnz], tf int32 [nnz], doc_len int32 [N]) and, for every doc id in
    `sample_docs`, that document's token list (the bench draws its query terms from documents)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w = torch.arange(1, vocab + 1, device=device, dtype=torch.float64).pow_(-s)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    del w
    lens_all = torch.poisson(torch.full((n_docs,), float(mean_len), device=device), generator=g).to(torch.int64)
    keys, sampled = [], {}
    want = None if sample_docs is None else torch.as_tensor(np.asarray(sample_docs, dtype=np.int64), device=device)
    TF_BITS = 10
    for d0 in range(0, n_docs, chunk_docs):
        d1 = min(n_docs, d0 + chunk_docs)
        lens = lens_all[d0:d1]
        total = int(lens.sum().item())
        u = torch.rand((total,), generator=g, device=device, dtype=torch.float64)
        tok = torch.searchsorted(cdf, u).clamp_(max=vocab - 1)
        del u
        doc = torch.repeat_interleave(torch.arange(d0, d1, device=device, dtype=torch.int64), lens)
        if want is not None:
            ptr = torch.cumsum(lens, 0) - lens
            for dd in want[(want >= d0) & (want < d1)].tolist():
                a = int(ptr[dd - d0].item())
                sampled[dd] = tok[a:a + int(lens[dd - d0].item())].cpu().numpy()
        k = torch.sort(tok * n_docs + doc).values
        del tok, doc
        uk, tf = torch.unique_consecutive(k, return_counts=True)
        del k
        keys.append((uk << TF_BITS) | tf.clamp_(max=(1 << TF_BITS) - 1))
        del uk, tf
    allk = torch.sort(torch.cat(keys)).values
    del keys
    tf = (allk & ((1 << TF_BITS) - 1)).to(torch.int32)
    allk >>= TF_BITS
    term = allk // n_docs
    doc = (allk - term * n_docs).to(torch.int32)
    del allk
    indptr = torch.zeros(vocab + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(torch.bincount(term, minlength=vocab), 0)
    del term
    out = (indptr.cpu().numpy(), doc.cpu().numpy(), tf.cpu().numpy(), lens_all.to(torch.int32).cpu().numpy())
    del indptr, doc, tf, lens_all, cdf
    torch.cuda.empty_cache()
    return out + (sampled,)


def term_queries_from_docs(sampled, order, seed=7):
    """4-12 tokens drawn (with replacement) from each sampled document, in `order` -> (term_ptr, terms) int32."""
    rng = np.random.default_rng(seed)
    ptr, terms = [0], []
    for dd in order:
        toks = sampled[int(dd)]
        n = int(rng.integers(4, 13))
        terms.extend(int(x) for x in (rng.choice(toks, n) if len(toks) else [0] * n))
        ptr.append(len(terms))
    return np.asarray(ptr, np.int32), np.asarray(terms, np.int32)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def timed_all_ranks(fn, steps, warmup, world):
    """bench.py contract: barrier + synchronize on both sides, MAX over ranks."""
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return float(dt.item()) / steps


def run_mode(args):
    import os
    import torch.distributed as dist
    from optimized_rag_amd import RagEngine
    from optimized_rag_amd.bm25 import Bm25Postings
    from optimized_rag_amd.sharded import ShardedHybridIndex, ShardedReranker, shard_bounds
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("RAG_BENCH_BACKEND", "nccl")       # gloo: rehearsal with all ranks on one GPU
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    device = torch.device("cuda", local)
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    eng = RagEngine(dim=DIM, device=local)
    out = {"mode": args.mode, "n_gpus": world, "data": "synthetic", "scaling": "strong",
           "steps": args.steps, "warmup": args.warmup}
    if args.mode == "hybrid":
        N, Q, k, pool = args.rows, args.queries, args.k, 100
        g = torch.Generator(device=device)
        g.manual_seed(1234)
        corpus = torch.randn((N, DIM), generator=g, device=device)        # same seed on every rank: replicated, then sliced
        corpus /= corpus.norm(dim=1, keepdim=True)
        rows = torch.randint(0, N, (Q,), generator=torch.Generator().manual_seed(4321))
        q = corpus[rows.to(device)] + torch.randn((Q, DIM), generator=g, device=device) * (0.5 / DIM ** 0.5)
        q = (q / q.norm(dim=1, keepdim=True)).contiguous()
        t0 = time.perf_counter()
        vocab = getattr(args, "vocab", 100_000)
        if vocab > 100_000:            # long-tailed vocabulary (millions of terms): generated on the GPU
            qdocs = np.random.default_rng(7).integers(0, N, Q)
            indptr, d, tf, dl, sampled = zipf_postings_gpu(N, vocab, 120, device, sample_docs=qdocs)
        else:
            indptr, d, tf, dl, tok, doc_ptr = synthetic_csr(N, vocab, 120)
        post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
        post.idf[np.diff(indptr) == 0] = 0.0
        sharded = None
        if world > 1:                  # rows, postings (global idf / avgdl) split contiguously
            b0, e0 = shard_bounds(N, world)[rank]
            sharded = ShardedHybridIndex(eng, rank=rank, world=world)
            sharded.load_shard(corpus[b0:e0].contiguous(), b0, post.shard(b0, e0))
        else:
            eng.index_load(corpus)
            post.load(eng)
        del corpus
        build_s = time.perf_counter() - t0
        if vocab > 100_000:
            ptr, terms = term_queries_from_docs(sampled, qdocs)
        else:
            ptr, terms = _term_queries(tok, doc_ptr, N, Q)                 # 4-12 tokens sampled from a random doc
        ptr_d, terms_d = torch.from_numpy(ptr).to(device), torch.from_numpy(terms).to(device)
        ids_d = torch.empty((Q, pool), dtype=torch.int64, device=device)
        sc_d = torch.empty((Q, pool), dtype=torch.float64, device=device)

        def dense():
            eng.dense_topk_dev(q, pool, ids_d, None, sc_d)

        def bm25():
            eng.bm25_topk_dev(ptr_d, terms_d, pool, ids_d, None, sc_d)

        def hybrid():
            if sharded is not None:
                return sharded.search_hybrid(q, ptr_d, terms_d, pool, k)
            return eng.hybrid_rrf_dev(q, ptr_d, terms_d, pool, k)

        only = getattr(args, "only_hybrid_calls", False)
        t_dense = 0.0 if only else timed(dense, args.steps, args.warmup)
        t_bm25 = 0.0 if only else timed(bm25, args.steps, 1)
        total = timed_all_ranks(hybrid, args.steps, 1, world)
        t_fuse = max(total - t_dense - t_bm25, 0.0)
        nnz_touched = float(np.diff(indptr)[terms[terms >= 0]].sum())
        from optimized_rag_amd._lib import bm25_index_bytes
        idx_bytes = bm25_index_bytes(indptr, N)
        out.update({
            "metric": "queries/sec (hybrid: dense top-100 + BM25 top-100 + RRF -> top-20)", "value": round(Q / total, 1),
            "unit": "queries/sec", "ms_per_step": round(total * 1e3, 3), "higher_is_better": True,
            "config": {"workload": f"{N} docs hybrid: dense + BM25(CSR, nnz={int(indptr[-1])}) + RRF(k=60), pool=100, top-k={k}, "
                                   f"batch={Q} (BASELINE.json configs[2])"},
            "stages_ms": {"dense_top100_dev": round(t_dense * 1e3, 3), "bm25_top100_dev": round(t_bm25 * 1e3, 3),
                          "rrf_fuse_dev (by difference)": round(t_fuse * 1e3, 3)},
            "bm25": {"postings_touched_per_batch": nnz_touched, "vocab": vocab, "distinct_terms": int((np.diff(indptr) > 0).sum()),
                     "algorithmic_GBs": None if only else round(nnz_touched * 12 / t_bm25 / 1e9, 1), "index_build_s": round(build_s, 1),
                     "index_bytes": {"postings": idx_bytes[0], "term_metadata": idx_bytes[1], "bracket_tables": idx_bytes[2]}},
            "note": "value = one rag_hybrid_rrf_dev call per batch, inputs and outputs resident in HBM" if sharded is None else
                    "value = local dense + BM25 lists, ONE all-gather (RCCL), two merges, RRF; max over ranks",
        })
    elif args.mode == "pipeline":
        # BASELINE.json configs[3]: hybrid top-100 -> cross-encoder rerank -> top-20, batch = 256 queries, ONE call
        from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors
        N, Q, k, pool, L, Ld, Lq = args.rows, min(args.queries, 256), args.k, 100, 256, 224, 16
        g = torch.Generator(device=device)
        g.manual_seed(1234)
        corpus = torch.randn((N, DIM), generator=g, device=device)
        corpus /= corpus.norm(dim=1, keepdim=True)
        rows = torch.randint(0, N, (Q,), generator=torch.Generator().manual_seed(4321))
        q = corpus[rows.to(device)] + torch.randn((Q, DIM), generator=g, device=device) * (0.5 / DIM ** 0.5)
        q = (q / q.norm(dim=1, keepdim=True)).contiguous()
        eng.index_load(corpus)
        del corpus
        indptr, d, tf, dl, tok, doc_ptr = synthetic_csr(N, 100_000, 120)
        post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
        post.idf[np.diff(indptr) == 0] = 0.0
        post.load(eng)
        rng = np.random.default_rng(7)
        ptr, terms = [0], []
        for i in range(Q):
            di = int(rng.integers(0, N))
            toks = tok[doc_ptr[di]:doc_ptr[di + 1]]
            n = int(rng.integers(4, 13))
            terms.extend(int(x) for x in (rng.choice(toks, n) if len(toks) else [0] * n))
            ptr.append(len(terms))
        ptr_d = torch.from_numpy(np.asarray(ptr, np.int32)).to(device)
        terms_d = torch.from_numpy(np.asarray(terms, np.int32)).to(device)
        cfg = MINILM_L6_CONFIG
        eng.ce_load(cfg, random_init_tensors(cfg, 2024))
        # passage token store: WordPiece ids ~U[1000, vocab), lengths ~U[96, 224], 16-token queries
        tok_store = torch.randint(1000, cfg["vocab_size"], (N, Ld), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
        tok_len = torch.randint(96, Ld + 1, (N,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
        eng.tokens_load(tok_store.numpy(), tok_len.numpy())
        del tok_store
        q_tok = torch.randint(1000, cfg["vocab_size"], (Q, Lq), generator=torch.Generator().manual_seed(8), dtype=torch.int32).to(device)
        q_len = torch.full((Q,), Lq, dtype=torch.int32, device=device)

        def run(nq):
            return eng.retrieve_rerank_dev(q[:nq], q_tok[:nq], q_len[:nq], pool, k, term_ptr=ptr_d[:nq + 1], terms=terms_d, L_pair=L)

        steps = max(2, args.steps // 5)
        t = timed(lambda: run(Q), steps, 1)
        lat = []
        for _ in range(5):
            torch.cuda.synchronize()
            a = time.perf_counter()
            run(Q)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - a)
        lat1 = []
        for it in range(12):
            torch.cuda.synchronize()
            a = time.perf_counter()
            run(1)
            torch.cuda.synchronize()
            if it >= 2:
                lat1.append(time.perf_counter() - a)
        ids, sc, lg, cand = run(Q)
        torch.cuda.synchronize()
        ok = bool((ids >= 0).all().item() and (sc[:, :-1] >= sc[:, 1:]).all().item())
        out.update({
            "metric": "queries/sec + p50 retrieve+rerank latency (hybrid top-100 -> cross-encoder -> top-20)",
            "value": round(Q / t, 2), "unit": "queries/sec", "steps": steps, "ms_per_step": round(t * 1e3, 2), "higher_is_better": True,
            "p50_batch_latency_ms": round(float(np.median(lat)) * 1e3, 2),
            "p50_single_query_latency_ms": round(float(np.median(lat1)) * 1e3, 3),
            "config": {"workload": f"{N} docs x {DIM}-d + BM25 CSR (nnz={int(indptr[-1])}) + {Ld}-token passage store; batch={Q} queries: "
                                   f"dense top-{pool} + BM25 top-{pool} + RRF -> top-{pool} -> MiniLM-L-6 cross-encoder (L={L}) -> top-{k} "
                                   f"(BASELINE.json configs[3]); one rag_retrieve_rerank_dev call per batch"},
            "sanity": {"all_slots_filled_and_sorted": ok},
        })
    else:
        from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors
        cfg = MINILM_L6_CONFIG
        eng.ce_load(cfg, random_init_tensors(cfg, 2024))
        Q, pool, L = min(args.queries, 256), 100, 256
        P = Q * pool
        rng = np.random.default_rng(5)
        lens = (16 + 2 + rng.integers(96, 225, P)).astype(np.int32).clip(max=L)
        ids = rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)
        tt = (np.arange(L)[None, :] >= 18).astype(np.int32).repeat(P, 0)
        ids_d = torch.from_numpy(ids).to(device)
        tt_d = torch.from_numpy(tt).to(device)
        lens_d = torch.from_numpy(lens).to(device)
        rer = ShardedReranker(eng, rank=rank, world=world)      # world == 1: a plain rag_ce_score_dev call

        def fwd():
            return rer.score(ids_d, tt_d, lens_d)

        out["steps"] = max(1, args.steps // 10)
        t = timed_all_ranks(fwd, out["steps"], 1, world)
        # SURVEY §8d per-pair formula 6 * len * (3.539e6 + 1536 * len), on the REAL token counts: padding is neither
        # computed (packed rows) nor counted
        flops = float((6.0 * lens.astype(np.float64) * (3.539e6 + 1536.0 * lens.astype(np.float64))).sum())
        out.update({
            "metric": "queries/sec (cross-encoder rerank of 100 candidates, L=256)", "value": round(Q / t, 2),
            "unit": "queries/sec", "ms_per_step": round(t * 1e3, 2), "higher_is_better": True,
            "config": {"workload": f"{Q} queries x {pool} pairs x L={L} tokens, ms-marco-MiniLM-L-6 shape, seeded weights "
                                   f"(BASELINE.json configs[3])"},
            "pairs_per_sec": round(P / t, 1),
            "roofline": {"bound": "mfma", "achieved": round(flops / t / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / t / 1e12 / 2500.0, 4),
                         "note": "algorithmic FLOPs of the real tokens (mean length %.0f of %d); the split-fp16 path issues 3 MFMAs per "
                                 "product, so the matrix pipe sees ~3x this rate" % (float(lens.mean()), L)},
        })
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# =====================================================================================================================
# Blocks of the DEFAULT bench line (python bench.py on one GPU): the rest of BASELINE.json's metric next to the dense
# headline, on the same resident 1M x 1536 index. Each returns a dict that carries its own roofline figures and a
# `cpu_baseline` timed on this box's host cores on a bounded sample (oracle/cpu_baseline.py).
# =====================================================================================================================
PEAK_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def _profile_number(name, path):
    """A number out of a committed profiles/ summary (the PMC passes cannot run inside the timed process); None if absent."""
    import os
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)) as f:
            d = json.load(f)
        for k in path:
            d = d[k]
        return float(d)
    except (OSError, ValueError, KeyError, TypeError):
        return None


def _term_queries(tok, doc_ptr, N, Q, seed=7):
    rng = np.random.default_rng(seed)
    ptr, terms = [0], []
    for i in range(Q):                                   # 4-12 tokens sampled from a random doc
        di = int(rng.integers(0, N))
        toks = tok[doc_ptr[di]:doc_ptr[di + 1]]
        n = int(rng.integers(4, 13))
        terms.extend(int(x) for x in (rng.choice(toks, n) if len(toks) else [0] * n))
        ptr.append(len(terms))
    return np.asarray(ptr, np.int32), np.asarray(terms, np.int32)


def _p50_ms(fn, n, warm):
    lat = []
    for it in range(warm + n):
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if it >= warm:
            lat.append((time.perf_counter() - a) * 1e3)
    return float(np.median(lat))


PEAK_L2_GBS = 34500.0            # aggregate L2 bandwidth, MI355X_MICROARCH.md (L2 per XCD)


def bm25_roofline(call_ms, algorithmic_bytes, bench_shape):
    """The three roofs of the BM25 top-100 leg (see the comment in hybrid_block). Counter figures come from the committed profile of
    the same command (they cannot be collected inside the timed run): the newest of profiles/r04_bm25_pmc.json, r03_bm25_pmc.json."""
    traffic, valu, src = None, None, None
    if bench_shape:
        for name in ("r04_bm25_pmc.json", "r03_bm25_pmc.json"):
            t = _profile_number(name, ("per_call", "hbm_traffic_bytes"))
            if t is not None:
                traffic, src = t, name
                valu = _profile_number(name, ("sq_counters_range_kernel", "valu_busy_per_simd"))
                break
    l2_gbs = algorithmic_bytes / (call_ms * 1e-3) / 1e9
    hbm_gbs = None if traffic is None else traffic / (call_ms * 1e-3) / 1e9
    return {"bound": "issue", "kernel": "bm25_plan_kernel + bm25_range_kernel + bm25_merge_select_kernel (BM25 top-100 of one batch)",
            "achieved": None if valu is None else round(valu, 4), "peak": 1.0, "unit": "vector-ALU busy share per SIMD (bm25_range_kernel)",
            "frac": None if valu is None else round(valu, 4),
            "hbm": {"achieved": None if hbm_gbs is None else round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": None if hbm_gbs is None else round(hbm_gbs / PEAK_HBM_GBS, 4), "bytes": "counter: left L2 per call"},
            "l2": {"achieved": round(l2_gbs, 1), "peak": PEAK_L2_GBS, "unit": "GB/s", "frac": round(l2_gbs / PEAK_L2_GBS, 4),
                   "bytes": "algorithmic: postings of the batch's query terms x 12 B"},
            "traffic": traffic, "traffic_source": None if src is None else f"profiles/{src} (bytes leaving L2 per 1024-query batch)",
            "avg_call_ms": round(call_ms, 4), "algorithmic_bytes_per_call": algorithmic_bytes,
            "note": "the posting stream is served by the L2s (hit rate 0.95): neither byte roof binds (HBM and L2 fractions above); the "
                    "scoring kernel is bound by instruction issue - `frac` is its vector-ALU busy share, with the scalar ALU at 0.59 and "
                    "waves parked 48 % of their lifetime beside it (profiles/r03_bm25_pmc.json)"}


def hybrid_block(eng, q, N, cpu_baseline=True):
    """BASELINE.json configs[2]: dense top-100 + BM25(CSR) top-100 + RRF(k=60) -> top-20, one rag_hybrid_rrf_dev call per
    batch. Loads the postings into `eng` (they stay resident for the retrieve_rerank block). Returns (block, state)."""
    from optimized_rag_amd.bm25 import Bm25Postings
    device = q.device
    Q, k, pool = q.shape[0], 20, 100
    t0 = time.perf_counter()
    indptr, d, tf, dl, tok, doc_ptr = synthetic_csr(N, 100_000, 120)
    post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
    post.idf[np.diff(indptr) == 0] = 0.0
    post.load(eng)
    build_s = time.perf_counter() - t0
    ptr, terms = _term_queries(tok, doc_ptr, N, Q)
    ptr_d, terms_d = torch.from_numpy(ptr).to(device), torch.from_numpy(terms).to(device)
    ids_d = torch.empty((Q, pool), dtype=torch.int64, device=device)
    sc_d = torch.empty((Q, pool), dtype=torch.float64, device=device)

    def hybrid(nq):
        return eng.hybrid_rrf_dev(q[:nq], ptr_d[:nq + 1], terms_d, pool, k)

    steps = 10
    t_1024 = timed(lambda: hybrid(Q), steps, 2)
    t_256 = timed(lambda: hybrid(256), steps, 2)
    # BM25 leg alone, device time from HIP events on the launch stream (rag_stage_kernel_ms, stage 1)
    eng.bm25_topk_dev(ptr_d, terms_d, pool, ids_d, None, sc_d)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    for _ in range(steps):
        eng.bm25_topk_dev(ptr_d, terms_d, pool, ids_d, None, sc_d)
    torch.cuda.synchronize()
    bm_ms, bm_spans = eng.stage_kernel_ms(1)
    eng.set_profiling(False)
    counts = np.diff(indptr)
    nnz_touched = float(counts[terms[terms >= 0]].sum())
    p50_1 = _p50_ms(lambda: hybrid(1), 100, 10)
    # the reference's OTHER fusion (rag/retrieval.py:294-322: weighted linear sum over every document), index-level
    t_lin = timed(lambda: eng.hybrid_linear_dev(q[:256], ptr_d[:257], terms_d, k, 0.55, 0.35, 0.10), 3, 1)
    block = {
        "workload": f"{N} docs: dense top-{pool} + BM25(CSR, nnz={int(indptr[-1])}) top-{pool} + RRF(k=60) -> top-{k} "
                    "(BASELINE.json configs[2]); one rag_hybrid_rrf_dev call per batch, everything resident in HBM",
        "value": round(Q / t_1024, 1), "unit": "queries/sec", "batch_queries": Q, "ms_per_batch": round(t_1024 * 1e3, 3),
        "queries_per_sec_batch256": round(256 / t_256, 1), "p50_single_query_latency_ms": round(p50_1, 4),
        "linear_fusion_queries_per_sec_batch256": round(256 / t_lin, 1),
        # BM25 is a gather over postings. SURVEY 8d prices it as algorithmic bytes (postings of the batch's query terms x 12 B) against
        # the HBM roof - but the queries of a batch share their frequent terms and the XCD-aware workgroup order lets each XCD's L2
        # serve a range's postings to a whole column of queries, so that quotient exceeds 1 (r3 reported frac 1.23) and binds nothing.
        # Reported instead, each against the roof it belongs to: (1) HBM: the bytes that actually LEFT L2 per batch (rocprofv3 --pmc
        # passes of `bench.py --mode hybrid --only-hybrid-calls`: FETCH_SIZE x 2 + WRITE_SIZE of the plan / range / merge launches)
        # over the measured device time; (2) L2: the algorithmic bytes over 34.5 TB/s (MI355X_MICROARCH.md, L2 section) - the roof
        # the posting stream is served from; (3) instruction issue: vector-ALU busy share of the scoring kernel (SQ counters).
        "roofline": bm25_roofline(bm_ms / bm_spans, nnz_touched * 12.0, N == 1_000_000 and Q == 1024),
        "index_build_s": round(build_s, 1),
    }
    if cpu_baseline:
        from oracle.cpu_baseline import bm25_topk_numpy
        ns = 24
        rows, cdt = bm25_topk_numpy(indptr, d, tf, dl, post.idf, post.avgdl, ptr[:ns + 1], terms, pool)
        ids_d2 = torch.empty((Q, pool), dtype=torch.int64, device=device)
        rows_d2 = torch.empty((Q, pool), dtype=torch.int32, device=device)
        eng.bm25_topk_dev(ptr_d, terms_d, pool, ids_d2, rows_d2, sc_d)
        torch.cuda.synchronize()
        same = float(np.mean([len(set(rows[i]) & set(rows_d2[i].cpu().numpy())) / pool for i in range(ns)]))
        block["cpu_baseline"] = {"value": round(ns / cdt, 2), "unit": "queries/sec (BM25 top-100 leg only)", "cores": 1, "kind": "port",
                                 "sample": f"{ns} of the {Q} term queries against the full {N}-doc postings: numpy BM25Okapi restatement "
                                           f"over CSR (all-doc float64 scores, /max, top-{pool}), {cdt:.2f}s",
                                 "topk_overlap_with_gpu": round(same, 5)}
    return block, dict(ptr_d=ptr_d, terms_d=terms_d, indptr=indptr)


def retrieve_rerank_block(eng, q, N, hyb_state, cpu_baseline=True):
    """BASELINE.json configs[3]: hybrid top-100 -> ms-marco-MiniLM-L-6 shape cross-encoder -> top-20, batch = 256 queries,
    ONE rag_retrieve_rerank_dev call per batch; p50 single-query retrieve+rerank latency over 100 calls."""
    from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors
    device = q.device
    Q, k, pool, L, Ld, Lq = 256, 20, 100, 256, 224, 16
    cfg = MINILM_L6_CONFIG
    tensors = random_init_tensors(cfg, 2024)
    eng.ce_load(cfg, tensors)
    # passage token store: WordPiece ids ~U[1000, vocab), lengths ~U[96, 224], 16-token queries
    tok_store = torch.randint(1000, cfg["vocab_size"], (N, Ld), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
    tok_len = torch.randint(96, Ld + 1, (N,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
    eng.tokens_load(tok_store.numpy(), tok_len.numpy())
    q_tok = torch.randint(1000, cfg["vocab_size"], (Q, Lq), generator=torch.Generator().manual_seed(8), dtype=torch.int32)
    q_tok_d = q_tok.to(device)
    q_len = torch.full((Q,), Lq, dtype=torch.int32, device=device)
    ptr_d, terms_d = hyb_state["ptr_d"], hyb_state["terms_d"]

    def run(nq):
        return eng.retrieve_rerank_dev(q[:nq], q_tok_d[:nq], q_len[:nq], pool, k, term_ptr=ptr_d[:nq + 1], terms=terms_d, L_pair=L)

    steps = 4
    run(Q)
    torch.cuda.synchronize()
    eng.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        run(Q)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / steps
    ce_ms, ce_spans = eng.stage_kernel_ms(2)
    eng.set_profiling(False)
    lat_b = _p50_ms(lambda: run(Q), 5, 0)
    lat_1 = _p50_ms(lambda: run(1), 100, 10)
    ids, sc, lg, cand = run(Q)
    torch.cuda.synchronize()
    cand_h = cand.cpu().numpy()
    ok = bool((ids >= 0).all().item() and (sc[:, :-1] >= sc[:, 1:]).all().item())
    # algorithmic FLOPs of one batch: SURVEY 8d per-pair formula 6 * len * (3.539e6 + 1536 * len) on the REAL token counts
    # of the pairs that were scored ([CLS] q [SEP] passage [SEP], longest_first to L); padding is neither computed nor counted
    plen = np.minimum(Lq + tok_len.numpy()[cand_h.reshape(-1)].astype(np.float64) + 3, L)
    flops = float((6.0 * plen * (3.539e6 + 1536.0 * plen)).sum())
    ce_tf = flops / (ce_ms / ce_spans * 1e-3) / 1e12
    block = {
        "workload": f"{N} docs x {DIM}-d + BM25 CSR + {Ld}-token passage store; batch={Q} queries: dense top-{pool} + BM25 top-{pool} "
                    f"+ RRF -> top-{pool} -> MiniLM-L-6 cross-encoder (L={L}, {Q * pool} pairs, mean {plen.mean():.0f} tokens) -> top-{k} "
                    "(BASELINE.json configs[3]); one rag_retrieve_rerank_dev call per batch",
        "value": round(Q / t, 2), "unit": "queries/sec", "batch_queries": Q, "ms_per_batch": round(t * 1e3, 2), "steps": steps,
        "p50_batch_latency_ms": round(lat_b, 2), "p50_single_query_latency_ms": round(lat_1, 3),
        "pairs_per_sec": round(Q * pool / t, 1),
        "roofline": {"bound": "mfma", "kernel": "cross-encoder forward (ce_gemm_kernel<*> + ce_attention_kernel + epilogues), all chunks of a batch",
                     "achieved": round(ce_tf, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ce_tf / PEAK_MFMA_TFLOPS, 4),
                     "traffic": _profile_number("r03_ce_traffic.json", ("per_forward_bytes", "total")) if N == 1_000_000 else None,
                     "traffic_source": "profiles/r03_ce_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE of one 25,600-pair forward; Infinity-Cache "
                                       "hits are counted in FETCH_SIZE)",
                     "avg_forward_ms": round(ce_ms / ce_spans, 3), "algorithmic_flops_per_forward": flops,
                     "share_of_batch_time": round(ce_ms / ce_spans / (t * 1e3), 4),
                     "note": "algorithmic FLOPs / device time of the forward (HIP events on the launch stream); operands are split-fp16 "
                             "(hi + lo) to hold the 1e-3 score bar, so the matrix pipe issues up to 3 MFMAs per algorithmic product"},
        "sanity": {"all_slots_filled_and_sorted": ok},
    }
    if cpu_baseline:
        from oracle.cpu_baseline import bert_cpu_pairs
        ns = 64                                              # 2 batches of 32 pairs = the first 64 pairs of query 0
        pid = np.zeros((ns, L), dtype=np.int64)
        ptt = np.zeros((ns, L), dtype=np.int64)
        pl = np.zeros(ns, dtype=np.int64)
        tok_np = tok_store.numpy()
        for j in range(ns):
            r = int(cand_h[0, j])
            dlen = int(min(tok_len[r], L - 3 - Lq))
            row = [101] + q_tok[0].tolist() + [102] + tok_np[r, :dlen].tolist() + [102]
            pid[j, :len(row)] = row
            ptt[j, Lq + 2:len(row)] = 1
            pl[j] = len(row)
        cl, cdt, threads = bert_cpu_pairs(cfg, tensors, pid, ptt, pl, batch=32)
        block["cpu_baseline"] = {"value": round(ns / cdt / pool, 4), "unit": "queries/sec (rerank of 100 pairs per query)", "cores": threads,
                                 "kind": "port", "pairs_per_sec": round(ns / cdt, 2),
                                 "sample": f"{ns} of the batch's {Q * pool} pairs (the first {ns} candidates of query 0), torch-CPU "
                                           f"BertForSequenceClassification fp32, batch 32 padded to the longest pair "
                                           f"(what sentence-transformers' CrossEncoder.predict does on CPU), {cdt:.2f}s"}
    del tok_store
    return block


class _Rows:
    """list-like payload table of the 1M-row index: rows are synthesised on access (no 1M dicts in memory)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {"content": f"chunk {int(i)}", "filename": "bench.txt", "file_type": "txt", "metadata": {}, "id": int(i)}


def agent_latency_block(eng, q, N, cpu_baseline=True):
    """What ONE agent turn calls, through the Python mirror classes (host pointers, synchronous C-ABI entries, dict assembly):
    GpuDocumentIndex.search (DocumentStore.search, rag/document_store.py:424-485, WHERE agent_id = ..., top_k 5),
    ConsistencyChecker.check_consistency on C = 60 claims (rag/consistency_checker.py:148-191), apply_mmr 12 -> 5
    (rag/nodes/helpers.py:183-260). p50 over 100 calls each, next to the reference's literal pure-Python loops."""
    from optimized_rag_amd.consistency_checker import ConsistencyChecker
    from optimized_rag_amd.document_store import GpuDocumentIndex
    from optimized_rag_amd.nodes_helpers import apply_mmr
    qh = q[:8].cpu().numpy()
    qlist = [[float(x) for x in row] for row in qh]

    class Svc:
        def __init__(self):
            self.i = 0

        def generate_embedding(self, text):
            self.i += 1
            return qlist[self.i % len(qlist)]

        def generate_embeddings_batch(self, texts):
            return [claim_emb[t] for t in texts]

    svc = Svc()
    idx = GpuDocumentIndex(svc, dim=DIM, engine=eng)
    idx.rows = _Rows(N)
    idx._tenant_id = {"agent-0": 0}
    eng.set_tenants(np.zeros(N, dtype=np.int32))

    def search():
        out = idx.search("agent-0", "what is in the corpus", top_k=5)
        assert len(out) == 5

    def wall_p50(fn, n=100, warm=10):
        lat = []
        for it in range(warm + n):
            a = time.perf_counter()
            fn()
            if it >= warm:
                lat.append((time.perf_counter() - a) * 1e3)
        return float(np.median(lat))

    p_search = wall_p50(search)
    search_raw = wall_p50(lambda: eng.dense_topk(qh[:1], 5, tenant=0))
    eng.set_tenants(None)
    # consistency: 5 documents x 12 claims of >= 20 characters, seeded claim embeddings with a few near-duplicates
    rng = np.random.default_rng(11)
    words = "memory vector index query document retrieval ranking fusion agent graph node embedding cosine score".split()
    docs, claim_emb, doc_of, texts = [], {}, [], []
    base = rng.standard_normal((60, DIM))
    base[7] = base[31] + 0.05 * rng.standard_normal(DIM)                          # one pair above the 0.85 threshold
    c = 0
    for di in range(5):
        sents = []
        for _ in range(12):
            t = " ".join(rng.choice(words, 7)) + f" number {c}"
            claim_emb[t] = [float(x) for x in base[c].astype(np.float32)]
            sents.append(t)
            texts.append(t)
            doc_of.append(di)
            c += 1
        docs.append({"content": ". ".join(sents) + ".", "source": f"doc_{di}"})
    chk = ConsistencyChecker(svc, similarity_threshold=0.85, engine=eng)
    res = chk.check_consistency([dict(x) for x in docs], "query")
    assert res.get("total_claims", 60) == 60, res
    p_cons = wall_p50(lambda: chk.check_consistency([dict(x) for x in docs], "query"))
    # apply_mmr: 12 retrieved documents carrying their embeddings -> 5
    cand_emb = rng.standard_normal((12, DIM)).astype(np.float32)
    mdocs = [{"content": f"d{i}", "embedding": [float(x) for x in cand_emb[i]]} for i in range(12)]
    p_mmr = wall_p50(lambda: apply_mmr("the query", mdocs, 0.7, 5, svc, engine=eng))
    block = {
        "workload": f"single calls through the Python mirror classes on the resident {N}-row index (host pointers in, dicts out)",
        "p50_ms": {"GpuDocumentIndex.search(agent_id, query, top_k=5) incl. 5 embedding fetches + dict assembly": round(p_search, 4),
                   "rag_dense_topk_host alone (Q=1, k=5, tenant filter)": round(search_raw, 4),
                   "ConsistencyChecker.check_consistency (5 docs, 60 claims, 1440 cross-document pairs)": round(p_cons, 4),
                   "apply_mmr (12 candidates -> 5, embeddings attached)": round(p_mmr, 4)},
        "calls_per_p50": 100,
    }
    if cpu_baseline:
        from oracle.cpu_baseline import python_loop_consistency, python_loop_mmr, python_loop_semantic_scan
        t_scan, _ = python_loop_semantic_scan(4096, DIM)
        t_cons, _ = python_loop_consistency([claim_emb[t] for t in texts], doc_of)
        t_mmr, _ = python_loop_mmr(qlist[1], cand_emb, 5, 0.7)
        block["cpu_baseline"] = {
            "kind": "port", "cores": 1, "unit": "ms per call",
            "value": {"semantic scan of ONE query over 4096 documents (rag/retrieval.py:253-256 loop; the reference's search itself "
                      "is a Postgres round trip, absent here)": round(t_scan * 1e3, 2),
                      "consistency pair loop, 60 claims (rag/consistency_checker.py:169-189)": round(t_cons * 1e3, 2),
                      "apply_mmr loop, 12 -> 5 (rag/nodes/helpers.py:229-252)": round(t_mmr * 1e3, 2)},
            "sample": "the literal pure-Python generator-sum cosine loops of the reference, one core, same vectors as the GPU calls "
                      "(N = 4096 for the scan: the full 1M-row loop would take minutes)"}
    return block
