"""Diagnostic (not product): the MX forward run N times on the same 7,680 mixed-length pairs must give the same bits every time (a race in
an epilogue - e.g. the LayerNorm statistics exchange, the staged parameters - would show as a run-to-run difference).
python tools/ce_determinism.py [runs]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd.cross_encoder import MINILM_L6_CONFIG, random_init_tensors  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = MINILM_L6_CONFIG
eng = RagEngine(dim=1536, device=0)
eng.ce_load(cfg, random_init_tensors(cfg, 2024))
rng = np.random.default_rng(11)
for P, L in ((7680, 256), (100, 256), (3000, 128), (513, 64)):
    lens = (18 + rng.integers(1, L - 17, P)).clip(max=L).astype(np.int32)
    ids = torch.from_numpy(rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)).cuda()
    tt = torch.zeros((P, L), dtype=torch.int32, device="cuda")
    ln = torch.from_numpy(lens).cuda()
    ref = None
    diff = 0
    for r in range(runs):
        out = torch.empty((P,), dtype=torch.float32, device="cuda")
        eng.ce_score_dev(ids, tt, ln, out)
        torch.cuda.synchronize()
        o = out.cpu().numpy().view(np.uint32)
        if ref is None:
            ref = o
        else:
            diff += int((o != ref).sum())
    print(f"P {P} L {L}: {runs} runs, logits differing from the first run: {diff}", flush=True)
