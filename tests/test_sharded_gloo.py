"""world_size-2 CPU test (gloo) of the multi-GPU host logic: row partitioning, the fused all-gather of
[ids | score bits] and the merge call sequence of optimized_rag_amd.sharded.ShardedDenseIndex.

No GPU here, so the per-rank engine is a TEST DOUBLE defined in this file (oracle-backed); the product class has no
CPU path of its own. The real HIP merge kernel is covered by tests/test_dense_gpu.py::test_merge_topk_equals_unsharded_search."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rag_oracle as O


class OracleEngine:
    """Stands in for RagEngine on CPU tensors: same method names / argument meaning."""

    def index_load(self, emb, ids=None, id_base=0):
        self.corpus = np.asarray(emb, dtype=np.float32)
        self.id_base = int(id_base)

    def dense_topk_dev(self, q, k, ids_out, rows_out, scores_out, tenant=-1, stream=None):
        ids, sc = O.dense_topk(self.corpus, q.numpy(), k)
        ids = np.where(ids >= 0, ids + self.id_base, -1)
        ids_out.copy_(torch.from_numpy(ids))
        scores_out.copy_(torch.from_numpy(sc))

    def merge_topk_dev(self, ids, scores, ids_out, scores_out, n_lists=None, list_stride=None, stream=None):
        Q, k = ids_out.shape
        # raw-pointer semantics of the C-ABI: list l starts l*list_stride elements after the first element handed in
        span = (n_lists - 1) * list_stride + Q * k
        flat = torch.as_strided(ids, (span,), (1,))
        flat_s = torch.as_strided(scores, (span,), (1,))
        for q in range(Q):
            cand = []
            for l in range(n_lists):
                for j in range(k):
                    i = int(flat[l * list_stride + q * k + j])
                    if i >= 0:
                        cand.append((-float(flat_s[l * list_stride + q * k + j]), i))
            cand.sort()
            for j in range(k):
                ids_out[q, j] = cand[j][1] if j < len(cand) else -1
                scores_out[q, j] = -cand[j][0] if j < len(cand) else 0.0


    # ---- BM25 / RRF / cross-encoder doubles for ShardedHybridIndex and ShardedReranker --------------------------
    def bm25_load(self, indptr, doc, tf, doc_len, idf, avgdl, k1=1.5, b=0.75):
        self.bm = (np.asarray(indptr), np.asarray(doc), np.asarray(tf), np.asarray(doc_len), np.asarray(idf), avgdl, k1, b)

    def bm25_set_normalize(self, on):
        self.bm_norm = bool(on)

    def bm25_topk_dev(self, term_ptr, terms, k, ids_out, rows_out, scores_out, raw_max_out=None, stream=None, tenant=-1):
        assert not self.bm_norm, "a row shard must hand out RAW scores"
        indptr, doc, tf, dl, idf, avgdl, k1, b = self.bm
        tp, tm = term_ptr.numpy(), terms.numpy()
        for q in range(tp.shape[0] - 1):
            sc = np.zeros(dl.shape[0])
            for t in tm[tp[q]:tp[q + 1]]:
                if t < 0:
                    continue
                a, e = int(indptr[t]), int(indptr[t + 1])
                d, f = doc[a:e], tf[a:e].astype(np.float64)
                sc[d] += idf[t] * (f * (k1 + 1) / (f + k1 * (1 - b + b * dl[d] / avgdl)))
            rows = O.stable_topk_desc(sc, k)
            for j in range(k):
                ids_out[q, j] = self.id_base + rows[j] if j < len(rows) else -1
                scores_out[q, j] = sc[rows[j]] if j < len(rows) else 0.0

    def rrf_fuse_dev(self, lists, keys_out, scores_out, ranks_out, rrf_k=60, stream=None):
        Q, top_k = keys_out.shape
        for q in range(Q):
            keys, sc, rk = O.rrf_fuse([[int(x) for x in l if x >= 0] for l in lists[q].tolist()], k=rrf_k, top_k=top_k)
            for j in range(top_k):
                keys_out[q, j] = keys[j] if j < len(keys) else -1
                scores_out[q, j] = sc[j] if j < len(keys) else 0.0
                ranks_out[q, j] = torch.tensor(rk[j] if j < len(keys) else [0] * lists.shape[1], dtype=torch.int32)

    def hybrid_fuse_gathered_dev(self, gathered, k, lists_out, scores_out, keys_out, rrf_out, ranks_out, rrf_k=60, stream=None):
        """Double of rag_hybrid_fuse_gathered_dev: two merges, BM25 / global max, RRF on the merged lists."""
        world, _, Q, pool = gathered.shape
        gf = gathered.view(torch.float64)
        stride = 4 * Q * pool
        self.merge_topk_dev(gathered, gf[:, 1], lists_out[0], scores_out[0], n_lists=world, list_stride=stride)
        self.merge_topk_dev(gathered[:, 2], gf[:, 3], lists_out[1], scores_out[1], n_lists=world, list_stride=stride)
        for q in range(Q):
            top = float(scores_out[1, q, 0])
            scores_out[1, q] /= top if top > 0 else 1.0
        self.rrf_fuse_dev(lists_out.permute(1, 0, 2), keys_out, rrf_out, ranks_out, rrf_k=rrf_k)

    def ce_score_dev(self, input_ids, token_type_ids, lens, logits_out, stream=None):
        logits_out.copy_(fake_logits(input_ids, token_type_ids, lens))

    def tokens_load(self, tokens, lens):
        self.tok, self.tok_len = np.asarray(tokens), np.asarray(lens)

    def ce_build_pairs_dev(self, q_tok, q_len, cand, ids_out, tt_out, lens_out, token_id_base=0, cls_id=101, sep_id=102, stream=None):
        Q, pool = cand.shape
        L = ids_out.shape[1]
        ids_out.zero_()
        tt_out.zero_()
        for q in range(Q):
            ql = int(min(int(q_len[q]), q_tok.shape[1], L - 3))
            for j in range(pool):
                r = int(cand[q, j]) - token_id_base if int(cand[q, j]) >= 0 else -1
                dl = 0 if r < 0 else int(min(self.tok_len[r], self.tok.shape[1], L - 3 - ql))
                row = [cls_id] + q_tok[q, :ql].tolist() + [sep_id] + ([] if r < 0 else self.tok[r, :dl].tolist()) + [sep_id]
                p = q * pool + j
                ids_out[p, :len(row)] = torch.tensor(row, dtype=torch.int32)
                tt_out[p, ql + 2:len(row)] = 1
                lens_out[p] = len(row)

    def rerank_topk_dev(self, logits, cand, ids_out, scores_out, logits_out, stream=None):
        Q, pool = cand.shape
        k = ids_out.shape[1]
        for q in range(Q):
            sc = [(O.sigmoid(float(logits[q * pool + j])), j) for j in range(pool) if int(cand[q, j]) >= 0]
            order = sorted(sc, key=lambda x: -x[0])
            for i in range(k):
                if i < len(order):
                    s_, j = order[i]
                    ids_out[q, i], scores_out[q, i], logits_out[q, i] = int(cand[q, j]), s_, float(logits[q * pool + j])
                else:
                    ids_out[q, i], scores_out[q, i], logits_out[q, i] = -1, 0.0, 0.0


def fake_logits(ids, tt, lens):
    """Any deterministic per-pair function: the reranker test checks the pair split / gather, not BERT."""
    return ((ids * (1 + tt)).sum(1) % 1000).to(torch.float32) / 100.0 - lens.to(torch.float32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimized_rag_amd.sharded import ShardedDenseIndex, shard_bounds
        rng = np.random.default_rng(5)
        D, Q, k = 64, 9, 7
        corpus = rng.standard_normal((n_rows, D)).astype(np.float32)
        corpus[3] = corpus[n_rows - 2]                       # exact tie across the two shards
        queries = (corpus[rng.integers(0, n_rows, Q)] + 0.3 * rng.standard_normal((Q, D))).astype(np.float32)
        queries[0] = corpus[3]
        b, e = shard_bounds(n_rows, world)[rank]
        idx = ShardedDenseIndex(OracleEngine(), rank=rank, world=world)
        idx.load_shard(corpus[b:e], b)
        ids, sc = idx.search(torch.from_numpy(queries), k)
        oid, osc = O.dense_topk(corpus, queries, k)
        ok = bool((ids.numpy() == oid).all() and np.abs(sc.numpy() - osc).max() < 1e-12)
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def _hybrid_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimized_rag_amd.bm25 import Bm25Postings
        from optimized_rag_amd.sharded import ShardedHybridIndex, ShardedReranker, shard_bounds
        rng = np.random.default_rng(11)
        N, D, Q, pool, k = 157, 32, 6, 12, 5
        words = [f"w{i}" for i in range(40)]
        corpus = [" ".join(rng.choice(words, size=int(rng.integers(3, 12)))) for _ in range(N)]
        emb = rng.standard_normal((N, D)).astype(np.float32)
        q = (emb[rng.integers(0, N, Q)] + 0.4 * rng.standard_normal((Q, D))).astype(np.float32)
        queries = [" ".join(rng.choice(words, size=3)) for _ in range(Q)]
        queries[2] = "not-in-vocabulary"
        post = Bm25Postings.from_corpus(corpus)
        ptr, terms = post.encode_queries(queries)
        b, e = shard_bounds(N, world)[rank]
        idx = ShardedHybridIndex(OracleEngine(), rank=rank, world=world)
        idx.load_shard(emb[b:e], b, post.shard(b, e))
        out = idx.search_hybrid(torch.from_numpy(q), torch.from_numpy(ptr), torch.from_numpy(terms), pool, k)
        # expectation: the unsharded oracle pipeline (dense top-pool, BM25 top-pool, RRF)
        d_rows, _ = O.dense_topk(emb, q, pool)
        obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
        ok = True
        for qi in range(Q):
            raw = obm.get_scores(O.tokenize(queries[qi]))
            b_rows = O.stable_topk_desc(raw, pool)
            keys, sc, rk = O.rrf_fuse([[int(r) for r in d_rows[qi]], [int(r) for r in b_rows]], k=60, top_k=k)
            mx = raw.max() if raw.max() > 0 else 1.0
            ok &= out["keys"][qi].tolist() == keys and out["rrf"][qi].tolist() == sc and out["ranks"][qi].tolist() == rk
            ok &= out["bm25_ids"][qi].tolist() == [int(r) for r in b_rows]
            ok &= out["bm25_scores"][qi].tolist() == [float(raw[r] / mx) for r in b_rows]
        # rerank split: P pairs not divisible by the world size
        P, L = 11, 16
        ids = torch.from_numpy(rng.integers(1000, 30000, (P, L)).astype(np.int32))
        tt = torch.from_numpy((np.arange(L)[None, :] >= 5).astype(np.int32).repeat(P, 0))
        lens = torch.from_numpy(rng.integers(6, L + 1, P).astype(np.int32))
        got = ShardedReranker(OracleEngine(), rank=rank, world=world).score(ids, tt, lens)
        ok &= bool(torch.equal(got, fake_logits(ids, tt, lens)))
        # the whole configs[4] composition: sharded hybrid candidates -> pairs from a replicated token store -> split rerank
        from optimized_rag_amd.sharded import ShardedPipeline
        Ld, Lq, Lp = 10, 4, 16
        tok = rng.integers(100, 999, (N, Ld)).astype(np.int32)
        tok_len = rng.integers(1, Ld + 1, N).astype(np.int32)
        q_tok = torch.from_numpy(rng.integers(100, 999, (Q, Lq)).astype(np.int32))
        q_len = torch.from_numpy(rng.integers(1, Lq + 1, Q).astype(np.int32))
        eng2 = OracleEngine()
        pipe = ShardedPipeline(eng2, rank=rank, world=world)
        pipe.index.load_shard(emb[b:e], b, post.shard(b, e))
        eng2.tokens_load(tok, tok_len)                                              # replicated store, ids from 0
        ids_k, sc_k, lg_k, cand = pipe.retrieve_rerank(torch.from_numpy(q), torch.from_numpy(ptr), torch.from_numpy(terms), q_tok, q_len,
                                                       pool, k, L_pair=Lp)
        whole = OracleEngine()                                                     # expectation: the same steps unsharded
        whole.tokens_load(tok, tok_len)
        want_cand = torch.full((Q, pool), -1, dtype=torch.int64)
        for qi in range(Q):
            raw = obm.get_scores(O.tokenize(queries[qi]))
            keys, _, _ = O.rrf_fuse([[int(r) for r in d_rows[qi]], [int(r) for r in O.stable_topk_desc(raw, pool)]], k=60, top_k=pool)
            want_cand[qi, :len(keys)] = torch.tensor(keys)
        pid = torch.zeros((Q * pool, Lp), dtype=torch.int32)
        ptt = torch.zeros((Q * pool, Lp), dtype=torch.int32)
        pln = torch.zeros((Q * pool,), dtype=torch.int32)
        whole.ce_build_pairs_dev(q_tok, q_len, want_cand, pid, ptt, pln)
        wi = torch.empty((Q, k), dtype=torch.int64)
        ws = torch.empty((Q, k), dtype=torch.float64)
        wl = torch.empty((Q, k), dtype=torch.float32)
        whole.rerank_topk_dev(fake_logits(pid, ptt, pln), want_cand, wi, ws, wl)
        ok &= bool(torch.equal(cand, want_cand) and torch.equal(ids_k, wi) and torch.equal(sc_k, ws) and torch.equal(lg_k, wl))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_hybrid_and_rerank_equal_global():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_hybrid_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("n_rows", [101, 40])
def test_two_rank_sharded_search_equals_global(n_rows):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_rows, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_bounds():
    from optimized_rag_amd.sharded import shard_bounds
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_bounds(1_000_000, 8)[-1] == (875_000, 1_000_000)
    assert shard_bounds(3, 8)[3] == (3, 3)
