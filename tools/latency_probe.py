"""Diagnostic: single-query hybrid / BM25 latency on the bench corpus (run under rocprofv3 --kernel-trace --stats to see
which launches the time goes to). Not part of the product or of bench.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_modes as BM  # noqa: E402
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.bm25 import Bm25Postings  # noqa: E402

N, D = int(os.environ.get("PROBE_ROWS", "1000000")), 1536
dev = torch.device("cuda", 0)
eng = RagEngine(dim=D, device=0)
g = torch.Generator(device=dev)
g.manual_seed(1)
eng.index_reserve(N)
for c in range(N // 125000):
    x = torch.randn((125000, D), generator=g, device=dev)
    eng.index_append(x)
q = torch.randn((8, D), generator=g, device=dev)
indptr, d, tf, dl, tok, doc_ptr = BM.synthetic_csr(N, 100_000, 120)
post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
post.idf[np.diff(indptr) == 0] = 0.0
post.load(eng)
ptr, terms = BM._term_queries(tok, doc_ptr, N, 8)
ptr_d, terms_d = torch.from_numpy(ptr).to(dev), torch.from_numpy(terms).to(dev)
ids = torch.empty((8, 100), dtype=torch.int64, device=dev)
sc = torch.empty((8, 100), dtype=torch.float64, device=dev)
for name, fn in (("warm-up (bm25 Q=8)", lambda: eng.bm25_topk_dev(ptr_d, terms_d, 100, ids, None, sc)),
                 ("bm25 Q=1", lambda: eng.bm25_topk_dev(ptr_d[:2], terms_d, 100, ids[:1], None, sc[:1])),
                 ("dense Q=1 k=100", lambda: eng.dense_topk_dev(q[:1], 100, ids[:1], None, sc[:1])),
                 ("hybrid Q=1", lambda: eng.hybrid_rrf_dev(q[:1], ptr_d[:2], terms_d, 100, 20)),
                 ("bm25 Q=8", lambda: eng.bm25_topk_dev(ptr_d, terms_d, 100, ids, None, sc))):
    print(name, "p50 ms", round(BM._p50_ms(fn, 30, 5), 4), flush=True)
