"""Diagnostic (not product): per-GEMM-site ablation of the split-fp16 correction terms of the cross-encoder.
For each RAG_CE_TERMS setting (one digit per site: qkv, out-proj, ffn-up, ffn-down; bit 0 = W_lo*x_hi, bit 1 = W_hi*x_lo)
score the same pairs, compare with the float64 BERT oracle (test infrastructure) and time a 4096-pair forward.
Writes a markdown table to stdout. The product library ships only the full-term kernels: this tool needs the ablation build,
  bash tools/ce_probe_build.sh ablation && RAG_HIP_LIB=tools/bin/librag_ablation.so python tools/ce_ablation.py > gpurun_out/ce_ablation.md"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.cross_encoder import flatten_state_dict  # noqa: E402
from oracle import bert_oracle as B  # noqa: E402

if "ablation" not in os.environ.get("RAG_HIP_LIB", ""):
    sys.exit("set RAG_HIP_LIB to tools/bin/librag_ablation.so (bash tools/ce_probe_build.sh ablation): RAG_CE_TERMS is not read by the product library")
cfg = B.minilm_config()
eng = RagEngine(dim=1536, device=0)
rng = np.random.default_rng(7)
P, L = 96, 256
lens = (18 + rng.integers(96, 225, P)).clip(max=L).astype(np.int32)
lens[:8] = [L, L, 5, 17, 64, 200, 33, 128]
results = {}
for seed in (99, 2024):                       # two seeded models: the test fixture's and the bench's
    w = B.seeded_weights(cfg, seed)
    eng.ce_load(cfg, flatten_state_dict(w, cfg["layers"]))
    ids = rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    tt = ((np.arange(L)[None, :] >= 18) & (np.arange(L)[None, :] < lens[:, None])).astype(np.int32)
    exp = B.forward_logits(w, cfg, ids.astype(np.int64), tt.astype(np.int64), lens, fast_erf=True)
    PB = 4096
    lb = (18 + rng.integers(96, 225, PB)).clip(max=L).astype(np.int32)
    idb = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (PB, L)).astype(np.int32)).cuda()
    ttb = torch.zeros((PB, L), dtype=torch.int32, device="cuda")
    lbd = torch.from_numpy(lb).cuda()
    out = torch.empty((PB,), dtype=torch.float32, device="cuda")
    configs = ["3333", "2333", "1333", "0333", "3233", "3133", "3033", "3323", "3313", "3303", "3332", "3331", "3330",
               "2222", "1111", "0000", "3322", "3321", "3312", "2332", "3223"]
    for c in configs:
        os.environ["RAG_CE_TERMS"] = c
        got = eng.ce_score(ids, tt, lens)
        err = got.astype(np.float64) - exp
        eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.ce_score_dev(idb, ttb, lbd, out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        results.setdefault(c, []).append((np.abs(err).max(), np.sqrt((err ** 2).mean()), ms))
    print(f"<!-- seed {seed}: logit range [{exp.min():.2f}, {exp.max():.2f}], std {exp.std():.2f} -->")
del os.environ["RAG_CE_TERMS"]
print("| RAG_CE_TERMS (qkv,out,ffn-up,ffn-down) | max abs logit err (seed 99 / 2024) | rms logit err | 4096-pair forward ms |")
print("|---|---|---|---|")
for c, r in results.items():
    print(f"| {c} | {r[0][0]:.2e} / {r[1][0]:.2e} | {r[0][1]:.2e} / {r[1][1]:.2e} | {r[0][2]:.1f} / {r[1][2]:.1f} |")
