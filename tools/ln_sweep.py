"""Diagnostic: forward time of the cross-encoder with the fused residual + LayerNorm GEMMs against the unfused form (GEMM with finer
tiles + a LayerNorm launch) over small batch sizes. Usage (GPU box): python tools/ln_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors  # noqa: E402

cfg = MINILM_L6_CONFIG
eng = RagEngine(dim=64, device=0)
eng.ce_load(cfg, random_init_tensors(cfg, 2024))
rng = np.random.default_rng(5)
L = 256
for P in (13, 25, 50, 100, 150, 200, 400, 800):
    lens = torch.from_numpy((16 + 2 + rng.integers(96, 225, P)).astype(np.int32).clip(max=L)).cuda()
    ids = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)).cuda()
    tt = torch.zeros((P, L), dtype=torch.int32, device="cuda")
    out = torch.empty((P,), dtype=torch.float32, device="cuda")
    res = {}
    for name, opt in (("fused", 0), ("unfused", 1), ("fused", 0), ("unfused", 1)):
        eng.set_option("ce_no_fused_ln", opt)
        for _ in range(3):
            eng.ce_score_dev(ids, tt, lens, out)
        torch.cuda.synchronize()
        n = max(5, 4000 // P)
        t0 = time.perf_counter()
        for _ in range(n):
            eng.ce_score_dev(ids, tt, lens, out)
        torch.cuda.synchronize()
        res.setdefault(name, []).append((time.perf_counter() - t0) / n * 1e3)
    print(f"pairs {P:5d} (~{int(lens.sum())} tokens): fused-LN {min(res['fused']):8.3f} ms   unfused {min(res['unfused']):8.3f} ms", flush=True)
