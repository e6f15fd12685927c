"""GPU: the row-sharded classes with the REAL library in TWO processes. The development boxes have one
GPU, so both ranks open their own handle on cuda:0 and exchange through gloo (the CPU test tests/test_sharded_gloo.py drives the
same classes with an oracle-backed engine double; an 8-GPU node runs them over RCCL). Every rank must end with exactly what the
one-call entries return on an unsharded index: dense top-k (ids, float64 scores), the hybrid result (RRF keys / scores / ranks,
merged lists, BM25 scores divided by the GLOBAL maximum - rag_hybrid_fuse_gathered_dev) and the configs[4] pipeline
(candidates, ids, sigmoid scores, logits) with the pairs split over the ranks and a replicated token store."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    from optimized_rag_amd.bm25 import Bm25Postings
    from oracle import bert_oracle as B
    rng = np.random.default_rng(2026)
    N, D, Q = 30_011, 256, 37                                # odd sizes: unequal shards, a partial last tile
    emb = rng.standard_normal((N, D)).astype(np.float32)
    emb[5] = emb[N - 3]                                      # an exact tie across the two shards
    q = (emb[rng.integers(0, N, Q)] + 0.4 * rng.standard_normal((Q, D))).astype(np.float32)
    q[0] = emb[5]
    words = [f"w{i}" for i in range(300)]
    corpus = [" ".join(rng.choice(words, size=int(rng.integers(3, 25)))) for _ in range(N)]
    post = Bm25Postings.from_corpus(corpus)
    queries = [" ".join(rng.choice(words, size=int(rng.integers(1, 6)))) for _ in range(Q)]
    queries[3] = "not-in-vocabulary"
    ptr, terms = post.encode_queries(queries)
    cfg = dict(vocab_size=2000, hidden=128, layers=2, heads=4, ffn=256, max_pos=64, type_vocab=2, eps=1e-12)
    w = B.seeded_weights(cfg, 17)
    Ld, Lq = 24, 6
    tok = rng.integers(5, 2000, (N, Ld)).astype(np.int32)
    tok_len = rng.integers(1, Ld + 1, N).astype(np.int32)
    q_tok = rng.integers(5, 2000, (Q, Lq)).astype(np.int32)
    q_len = rng.integers(1, Lq + 1, Q).astype(np.int32)
    return dict(N=N, D=D, Q=Q, emb=emb, q=q, post=post, ptr=ptr, terms=terms, cfg=cfg, w=w, tok=tok, tok_len=tok_len, q_tok=q_tok, q_len=q_len)


def _rank(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optimized_rag_amd import RagEngine
        from optimized_rag_amd.cross_encoder import flatten_state_dict
        from optimized_rag_amd.sharded import ShardedDenseIndex, ShardedPipeline, shard_bounds
        d = _data()
        dev = torch.device("cuda", 0)
        pool, k, Lp = 50, 10, 32
        cuda = lambda a: torch.from_numpy(a).to(dev)
        q, ptr, terms, q_tok, q_len = cuda(d["q"]), cuda(d["ptr"]), cuda(d["terms"]), cuda(d["q_tok"]), cuda(d["q_len"])
        b, e = shard_bounds(d["N"], world)[rank]
        eng = RagEngine(dim=d["D"], device=0)
        pipe = ShardedPipeline(eng, rank=rank, world=world)
        pipe.index.load_shard(d["emb"][b:e], b, d["post"].shard(b, e))
        eng.ce_load(d["cfg"], flatten_state_dict(d["w"], d["cfg"]["layers"]))
        eng.tokens_load(d["tok"], d["tok_len"])                                      # REPLICATED store, ids from 0
        dense = ShardedDenseIndex(eng, rank=rank, world=world)
        ids, sc = dense.search(q, k)
        hyb = pipe.index.search_hybrid(q, ptr, terms, pool, k)
        r_ids, r_sc, r_lg, r_cand = pipe.retrieve_rerank(q, ptr, terms, q_tok, q_len, pool, k, L_pair=Lp)
        torch.cuda.synchronize()
        got = dict(ids=ids.cpu(), sc=sc.cpu(), **{kk: v.cpu().clone() for kk, v in hyb.items()}, r_ids=r_ids.cpu(), r_sc=r_sc.cpu(),
                   r_lg=r_lg.cpu(), r_cand=r_cand.cpu())
        ok = True
        if rank == 0:                                         # the unsharded truth, same library, one-call entries
            whole = RagEngine(dim=d["D"], device=0)
            whole.index_load(d["emb"])
            d["post"].load(whole)
            whole.ce_load(d["cfg"], flatten_state_dict(d["w"], d["cfg"]["layers"]))
            whole.tokens_load(d["tok"], d["tok_len"])
            wi = torch.empty((d["Q"], k), dtype=torch.int64, device=dev)
            ws = torch.empty((d["Q"], k), dtype=torch.float64, device=dev)
            whole.dense_topk_dev(q, k, wi, None, ws)
            keys, rrf, ranks = whole.hybrid_rrf_dev(q, ptr, terms, pool, k)
            bi = torch.empty((d["Q"], pool), dtype=torch.int64, device=dev)
            bs = torch.empty((d["Q"], pool), dtype=torch.float64, device=dev)
            whole.bm25_topk_dev(ptr, terms, pool, bi, None, bs)
            torch.cuda.synchronize()
            ok &= torch.equal(got["ids"], wi.cpu()) and torch.equal(got["sc"], ws.cpu())
            ok &= torch.equal(got["keys"], keys.cpu()) and torch.equal(got["rrf"], rrf.cpu()) and torch.equal(got["ranks"], ranks.cpu())
            ok &= torch.equal(got["bm25_ids"], bi.cpu()) and torch.equal(got["bm25_scores"], bs.cpu())       # / GLOBAL max, bit for bit
            o_ids, o_sc, o_lg, o_cand = whole.retrieve_rerank_dev(q, q_tok, q_len, pool, k, term_ptr=ptr, terms=terms, L_pair=Lp)
            torch.cuda.synchronize()
            ok &= torch.equal(got["r_cand"], o_cand.cpu()) and torch.equal(got["r_ids"], o_ids.cpu())
            ok &= torch.equal(got["r_lg"], o_lg.cpu()) and torch.equal(got["r_sc"], o_sc.cpu())
            whole.close()
        # every rank holds the same result: compare with rank 0's
        for name in ("ids", "keys", "r_ids", "r_cand"):
            ref = got[name].clone()
            dist.broadcast(ref, src=0)
            ok &= torch.equal(ref, got[name])
        ret[rank] = bool(ok)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_on_one_gpu_equal_the_unsharded_entries():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
