import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench as BE, bench_shard as BS, bench_modes as BM
from optimized_rag_amd import RagEngine
rows = int(sys.argv[1])
dev = torch.device("cuda", 0)
eng = RagEngine(dim=BE.DIM, device=0)
st = BS.build_shard(eng, dev, rows, Q=256, with_rerank=False, log=lambda *a: None)
q = st["queries"]
ids = torch.empty((256, 100), dtype=torch.int64, device=dev); sc = torch.empty((256, 100), dtype=torch.float64, device=dev); rw = torch.empty((256, 100), dtype=torch.int32, device=dev)
for flag in (0, 1, 0, 1):
    eng.set_option("bm25_sort_merge", flag)
    t = BM.timed_all_ranks(lambda: eng.bm25_topk_dev(st["ptr_d"], st["terms_d"], 100, ids, rw, sc), 3, 1, 1)
    th = BM.timed_all_ranks(lambda: eng.hybrid_rrf_dev(q, st["ptr_d"], st["terms_d"], 100, 20), 3, 1, 1)
    print(f"rows {rows} sort_merge={flag}: bm25 top-100 {t*1e3:.3f} ms, hybrid {th*1e3:.3f} ms", flush=True)
