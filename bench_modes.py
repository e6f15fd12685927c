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
        indptr, d, tf, dl, tok, doc_ptr = synthetic_csr(N, 100_000, 120)
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
        rng = np.random.default_rng(7)
        ptr, terms = [0], []
        for i in range(Q):                                   # 4-12 tokens sampled from a random doc
            di = int(rng.integers(0, N))
            toks = tok[doc_ptr[di]:doc_ptr[di + 1]]
            n = int(rng.integers(4, 13))
            terms.extend(int(x) for x in (rng.choice(toks, n) if len(toks) else [0] * n))
            ptr.append(len(terms))
        ptr, terms = np.asarray(ptr, np.int32), np.asarray(terms, np.int32)
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

        t_dense = timed(dense, args.steps, args.warmup)
        t_bm25 = timed(bm25, args.steps, 1)
        total = timed_all_ranks(hybrid, args.steps, 1, world)
        t_fuse = max(total - t_dense - t_bm25, 0.0)
        nnz_touched = float(sum(int(indptr[t + 1] - indptr[t]) for t in terms if t >= 0))
        out.update({
            "metric": "queries/sec (hybrid: dense top-100 + BM25 top-100 + RRF -> top-20)", "value": round(Q / total, 1),
            "unit": "queries/sec", "ms_per_step": round(total * 1e3, 3), "higher_is_better": True,
            "config": {"workload": f"{N} docs hybrid: dense + BM25(CSR, nnz={int(indptr[-1])}) + RRF(k=60), pool=100, top-k={k}, "
                                   f"batch={Q} (BASELINE.json configs[2])"},
            "stages_ms": {"dense_top100_dev": round(t_dense * 1e3, 3), "bm25_top100_dev": round(t_bm25 * 1e3, 3),
                          "rrf_fuse_dev (by difference)": round(t_fuse * 1e3, 3)},
            "bm25": {"postings_touched_per_batch": nnz_touched,
                     "algorithmic_GBs": round(nnz_touched * 12 / t_bm25 / 1e9, 1), "index_build_s": round(build_s, 1)},
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
