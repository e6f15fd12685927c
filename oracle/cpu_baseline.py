"""TEST / MEASUREMENT INFRASTRUCTURE — CPU baseline leg of bench.py ("port" kind). Not product code.

The reference's dense retrieval is `ORDER BY embedding <=> q LIMIT k` inside Postgres/pgvector
(/root/reference/rag/document_store.py:448-460); neither Postgres nor the extension exists on the GPU box, so
the CPU baseline is this restatement of the exact scan: float32 Q.C^T on unit rows with the host BLAS on all
cores + argpartition/sort. Same semantics as oracle/rag_oracle.py::dense_topk, traded
float64 row loops for BLAS so that a bounded sample finishes in seconds.
"""
import os
import time

import numpy as np


def effective_cores():
    """CPU threads this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def dense_topk_blas(corpus_unit, queries_unit, k, threads=None):
    """corpus_unit [N,D] float32 unit rows (torch CPU tensor), queries_unit [Q,D]. Returns (idx [Q,k], seconds)."""
    import torch
    threads = threads or effective_cores()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    s = queries_unit @ corpus_unit.T                        # [Q,N] float32, BLAS on all cores
    top = torch.topk(s, k, dim=1, largest=True, sorted=True)
    dt = time.perf_counter() - t0
    return top.indices.numpy(), top.values.numpy(), dt, threads


def python_loop_cosine_rate(dim=1536, n=2000):
    """The literal reference loop (rag/retrieval.py:253-256,362-371): pure-Python cosine, one core. pairs/s."""
    import math
    rng = np.random.default_rng(0)
    q = [float(x) for x in rng.standard_normal(dim)]
    docs = [[float(x) for x in rng.standard_normal(dim)] for _ in range(n)]
    t0 = time.perf_counter()
    for d in docs:
        dot = sum(a * b for a, b in zip(q, d))
        m1 = math.sqrt(sum(a * a for a in q))
        m2 = math.sqrt(sum(b * b for b in d))
        _ = dot / (m1 * m2)
    return n / (time.perf_counter() - t0)
