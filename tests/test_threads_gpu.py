"""The engine handle is shared by the agent's worker threads (the reference's connection pool allows 10 concurrent
`DocumentStore.search` calls, /root/reference/database/connection.py:38-42): every C-ABI entry takes the handle's mutex, so
concurrent host-pointer calls on ONE handle must return exactly what the same calls return one after the other (ctypes
releases the GIL during a call, so the threads really are inside the library at the same time)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_concurrent_host_calls_on_one_handle_match_the_serial_results():
    from optimized_rag_amd import RagEngine
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(10)
    N, D, T = 30000, 1536, 8
    emb = rng.standard_normal((N, D)).astype(np.float32)
    eng = RagEngine(dim=D, device=0)
    try:
        eng.index_load(emb)
        docs = [" ".join(f"t{t}" for t in rng.integers(0, 800, 20)) for _ in range(N)]
        post = Bm25Postings.from_corpus(docs).load(eng)
        work = []
        for t in range(T):
            q = (emb[rng.integers(0, N, 3)] + 0.3 * rng.standard_normal((3, D))).astype(np.float32)
            ptr, terms = post.encode_queries([docs[int(rng.integers(0, N))] for _ in range(3)])
            a = rng.standard_normal((5 + t, D)).astype(np.float32)
            work.append((q, ptr, terms, a))

        def calls(item):
            q, ptr, terms, a = item
            ids, rows, sc = eng.dense_topk(q, 7 + len(a) % 3)
            b_ids, _, b_sc, _ = eng.bm25_topk(ptr, terms, 10)
            return ids.copy(), sc.copy(), b_ids.copy(), b_sc.copy(), eng.pairwise_cosine(a).copy()

        serial = [calls(w) for w in work]
        out, errs = [None] * T, []

        def worker(i):
            try:
                for _ in range(6):                      # several rounds: the threads keep colliding inside the library
                    out[i] = calls(work[i])
            except Exception as e:                       # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for got, exp in zip(out, serial):
            for g, e in zip(got, exp):
                np.testing.assert_array_equal(g, e)
    finally:
        eng.close()
