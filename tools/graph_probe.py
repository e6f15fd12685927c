"""Diagnostic: p50 of ONE-query hybrid and retrieve+rerank calls issued directly vs replayed from a captured graph
(torch.cuda.CUDAGraph around the same C-ABI calls on the capturing stream). Not part of the product or of bench.py."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_modes as BM  # noqa: E402
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.bm25 import Bm25Postings  # noqa: E402
from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors  # noqa: E402

N, D = 1_000_000, 1536
dev = torch.device("cuda", 0)
eng = RagEngine(dim=D, device=0)
g = torch.Generator(device=dev)
g.manual_seed(1)
eng.index_reserve(N)
for c in range(N // 125000):
    eng.index_append(torch.randn((125000, D), generator=g, device=dev))
q = torch.randn((8, D), generator=g, device=dev)
indptr, d, tf, dl, tok, doc_ptr = BM.synthetic_csr(N, 100_000, 120)
post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
post.idf[np.diff(indptr) == 0] = 0.0
post.load(eng)
ptr, terms = BM._term_queries(tok, doc_ptr, N, 8)
ptr_d, terms_d = torch.from_numpy(ptr).to(dev), torch.from_numpy(terms).to(dev)
cfg = MINILM_L6_CONFIG
eng.ce_load(cfg, random_init_tensors(cfg, 2024))
Ld, Lq, L = 224, 16, 256
tok_store = torch.randint(1000, cfg["vocab_size"], (N, Ld), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
tok_len = torch.randint(96, Ld + 1, (N,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
eng.tokens_load(tok_store.numpy(), tok_len.numpy())
q_tok = torch.randint(1000, cfg["vocab_size"], (8, Lq), generator=torch.Generator().manual_seed(8), dtype=torch.int32).to(dev)
q_len = torch.full((8,), Lq, dtype=torch.int32, device=dev)


def p50(fn, n=100, warm=10):
    lat = []
    for it in range(n + warm):
        torch.cuda.synchronize()
        a = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if it >= warm:
            lat.append((time.perf_counter() - a) * 1e3)
    return round(float(np.median(lat)), 4)


cases = {
    "hybrid Q=1": lambda: eng.hybrid_rrf_dev(q[:1], ptr_d[:2], terms_d, 100, 20),
    "retrieve_rerank Q=1": lambda: eng.retrieve_rerank_dev(q[:1], q_tok[:1], q_len[:1], 100, 20, term_ptr=ptr_d[:2], terms=terms_d, L_pair=L),
}
side = torch.cuda.Stream()
for name, fn in cases.items():
    direct = p50(fn)
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(gr, stream=side):
                out = fn()
            replay = p50(gr.replay)
        except Exception as e:  # noqa: BLE001
            replay = f"capture failed: {e!r}"[:200]
    print(name, "direct p50 ms", direct, "| graph replay p50 ms", replay, flush=True)
