"""Diagnostic (not product): logit error and forward time of the MX forward (option ce_mx) against the split-fp16 forward and the
float64 BERT oracle (test infrastructure), on the pairs of tools/ce_ablation.py.  python tools/ce_mx_check.py [pairs_timed]"""
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
rng = np.random.default_rng(7)
P, L = 192, 256
PB = int(sys.argv[1]) if len(sys.argv) > 1 else 7680
lens = (18 + rng.integers(96, 225, P)).clip(max=L).astype(np.int32)
lens[:8] = [L, L, 5, 17, 64, 200, 33, 128]
for seed in (99, 2024):
    w = B.seeded_weights(cfg, seed)
    eng.ce_load(cfg, flatten_state_dict(w, cfg["layers"]))
    ids = rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    tt = ((np.arange(L)[None, :] >= 18) & (np.arange(L)[None, :] < lens[:, None])).astype(np.int32)
    exp = B.forward_logits(w, cfg, ids.astype(np.int64), tt.astype(np.int64), lens, fast_erf=True)
    lb = (18 + rng.integers(96, 225, PB)).clip(max=L).astype(np.int32)
    idb = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (PB, L)).astype(np.int32)).cuda()
    ttb = torch.zeros((PB, L), dtype=torch.int32, device="cuda")
    lbd = torch.from_numpy(lb).cuda()
    out = torch.empty((PB,), dtype=torch.float32, device="cuda")
    for mx in (-1, 1):
        eng.set_option("ce_mx", mx)
        got = eng.ce_score(ids, tt, lens)
        err = got.astype(np.float64) - exp
        eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"seed {seed} ce_mx {mx:2d}: max |logit err| {np.abs(err).max():.3e} rms {np.sqrt((err ** 2).mean()):.3e} | {PB} pairs {ms:.1f} ms = {PB / ms * 1e3:.0f} pairs/s"
              f" | logits [{exp.min():.2f}, {exp.max():.2f}]", flush=True)
    eng.set_option("ce_mx", 0)
