"""Diagnostic: do two dense searches issued on two streams (two handles, same corpus) overlap their latency-bound small
stages with each other's big GEMM stage?  python tools/overlap_probe.py [rows] [queries]"""
import sys, time
import torch
sys.path.insert(0, ".")
from optimized_rag_amd import RagEngine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
corpus = torch.randn((rows, 1536), generator=g, device=dev)
q = (corpus[torch.randint(0, rows, (Q,), device=dev)] + 0.5 * torch.randn((Q, 1536), generator=g, device=dev)).contiguous()
engs = [RagEngine(dim=1536, device=0) for _ in range(2)]
for e in engs:
    e.index_load(corpus)
del corpus
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
outs = [(torch.empty((Q, 20), dtype=torch.int64, device=dev), torch.empty((Q, 20), dtype=torch.float64, device=dev)) for _ in range(2)]

def run(n_slots, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % n_slots
        with torch.cuda.stream(streams[s]):
            engs[s].dense_topk_dev(q, 20, outs[s][0], None, outs[s][1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for n in (1, 2):
    run(n, 6)
    print(f"rows={rows} Q={Q} slots={n}: {run(n, 40):.4f} ms per batch", flush=True)
assert torch.equal(outs[0][0], outs[1][0])
