"""Diagnostic (not product): forward time of the MX forward (option ce_mx = 1) against the split-fp16 forward (ce_mx = -1) over batch
sizes - where MX_MIN_ROWS (cross_encoder.hip) should sit.  python tools/ce_mx_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.cross_encoder import flatten_state_dict  # noqa: E402
from oracle import bert_oracle as B  # noqa: E402

cfg = B.minilm_config()
eng = RagEngine(dim=1536, device=0)
eng.ce_load(cfg, flatten_state_dict(B.seeded_weights(cfg, 99), cfg["layers"]))
rng = np.random.default_rng(7)
L = 256
print("pairs | split-fp16 ms | MX ms")
for P in (13, 25, 50, 100, 150, 200, 300, 400, 600, 800, 1200, 1600, 3200, 6400):
    lb = (18 + rng.integers(96, 225, P)).clip(max=L).astype(np.int32)
    idb = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)).cuda()
    ttb = torch.zeros((P, L), dtype=torch.int32, device="cuda")
    lbd = torch.from_numpy(lb).cuda()
    out = torch.empty((P,), dtype=torch.float32, device="cuda")
    res = []
    for mx in (-1, 1):
        eng.set_option("ce_mx", mx)
        for _ in range(3):
            eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        n = 20 if P <= 800 else 5
        t0 = time.perf_counter()
        for _ in range(n):
            eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
    print(f"{P:6d} | {res[0]:8.3f} | {res[1]:8.3f}", flush=True)
