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
        fi = ids.reshape(-1)
        fs = scores.reshape(-1) if scores.is_contiguous() else None
        base_s = (scores.data_ptr() - ids.data_ptr()) // 8
        flat = ids.reshape(-1)
        flat_s = flat.view(torch.float64)
        for q in range(Q):
            cand = []
            for l in range(n_lists):
                for j in range(k):
                    i = int(flat[l * list_stride + q * k + j])
                    if i >= 0:
                        cand.append((-float(flat_s[base_s + l * list_stride + q * k + j]), i))
            cand.sort()
            for j in range(k):
                ids_out[q, j] = cand[j][1] if j < len(cand) else -1
                scores_out[q, j] = -cand[j][0] if j < len(cand) else 0.0


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
