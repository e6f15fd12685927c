"""Diagnostic: 6 index-level linear-fusion calls of 256 queries on the 1M-document bench index (run under rocprofv3 --kernel-trace
--stats for the per-kernel split of rag_hybrid_linear_dev)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BE  # noqa: E402
import bench_modes as BM  # noqa: E402
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.bm25 import Bm25Postings  # noqa: E402

N = 1_000_000
dev = torch.device("cuda", 0)
eng = RagEngine(dim=BE.DIM, device=0)
eng.index_reserve(N)
for c in range(N // BE.CHUNK_ROWS):
    eng.index_append(BE.gen_chunk(c, BE.CHUNK_ROWS, dev))
q, _ = BE.gen_queries(256, N, N // BE.CHUNK_ROWS, BE.CHUNK_ROWS, dev)
indptr, d, tf, dl, tok, doc_ptr = BM.synthetic_csr(N, 100_000, 120)
post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
post.idf[np.diff(indptr) == 0] = 0.0
post.load(eng)
ptr, terms = BM._term_queries(tok, doc_ptr, N, 256)
ptr_d, terms_d = torch.from_numpy(ptr).to(dev), torch.from_numpy(terms).to(dev)
for _ in range(2):
    eng.hybrid_linear_dev(q, ptr_d, terms_d, 20, 0.55, 0.35, 0.10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    eng.hybrid_linear_dev(q, ptr_d, terms_d, 20, 0.55, 0.35, 0.10)
torch.cuda.synchronize()
print("ms per 256-query call:", (time.perf_counter() - t0) / 6 * 1e3)
