"""Diagnostic: forward time of one 25,600-pair batch (L = 256, mixed lengths) against the activation chunk size (option ce_chunk_tokens).
python tools/ce_chunk_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors  # noqa: E402

cfg = MINILM_L6_CONFIG
eng = RagEngine(dim=1536, device=0)
eng.ce_load(cfg, random_init_tensors(cfg, 2024))
rng = np.random.default_rng(3)
P, L = 25600, 256
lens = (18 + rng.integers(96, 225, P)).clip(max=L).astype(np.int32)
ids = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)).cuda()
tt = torch.zeros((P, L), dtype=torch.int32, device="cuda")
ln = torch.from_numpy(lens).cuda()
out = torch.empty((P,), dtype=torch.float32, device="cuda")
for ct in (1_000_000, 2_000_000, 3_400_000, 7_000_000, 2_000_000):
    eng.set_option("ce_chunk_tokens", ct)
    eng.ce_score_dev(ids, tt, ln, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.ce_score_dev(ids, tt, ln, out)
    torch.cuda.synchronize()
    print(f"ce_chunk_tokens {ct}: {(time.perf_counter() - t0) / 4 * 1e3:.1f} ms per 25,600 pairs", flush=True)
