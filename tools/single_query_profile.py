"""Diagnostic: 60 single-query retrieve+rerank calls (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os
import sys

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
tok_store = torch.randint(1000, cfg["vocab_size"], (N, 224), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
tok_len = torch.randint(96, 225, (N,), generator=torch.Generator().manual_seed(6), dtype=torch.int32)
eng.tokens_load(tok_store.numpy(), tok_len.numpy())
q_tok = torch.randint(1000, cfg["vocab_size"], (8, 16), generator=torch.Generator().manual_seed(8), dtype=torch.int32).to(dev)
q_len = torch.full((8,), 16, dtype=torch.int32, device=dev)
for _ in range(60):
    eng.retrieve_rerank_dev(q[:1], q_tok[:1], q_len[:1], 100, 20, term_ptr=ptr_d[:2], terms=terms_d, L_pair=256)
torch.cuda.synchronize()
