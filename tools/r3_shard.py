"""The per-GPU share of BASELINE.json configs[4] on ONE MI355X, measured: 12.5M x 1536 rows + ~1.2e9 postings
over a 2M-term vocabulary + passage token store + cross-encoder. Prints one JSON document:
  * index sizes (postings / term metadata / bracket tables, HBM in use),
  * dense (1024 / 256 / 128 queries), hybrid (256) and one-call retrieve + rerank (256) rates,
  * the same retrieve + rerank with the token store REPLICATED for an 8-GPU node (100M x 224 passages = 44.8 GB resident) through
    ShardedPipeline(world = 1), its result compared with the one-call entry, and the peak HBM figure of that configuration.
Parity against the oracle at this size is tests/test_shard_share_gpu.py with RAG_TEST_SHARD_ROWS=12500000 (tools/r3_shard.sh runs both).
Usage: python tools/r3_shard.py [rows_per_gpu] > gpurun_out/r3_shard.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BE  # noqa: E402
import bench_shard as BS  # noqa: E402
from optimized_rag_amd import RagEngine  # noqa: E402
from optimized_rag_amd._lib import bm25_index_bytes  # noqa: E402
from optimized_rag_amd.sharded import ShardedPipeline  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
replicate_world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
eng = RagEngine(dim=BE.DIM, device=0)
log = lambda *a: print(*a, file=sys.stderr, flush=True)
t0 = time.perf_counter()
st = BS.build_shard(eng, dev, rows, Q=1024, log=log)
build_s = time.perf_counter() - t0
post = st["post"]
pb, mb, tb = bm25_index_bytes(post.indptr, rows)
df = np.diff(post.indptr)
out = {"rows_per_gpu": rows, "dim": BE.DIM, "build_s": round(build_s, 1),
       "bm25_index": {"nnz": int(post.indptr[-1]), "vocab": int(df.shape[0]), "distinct_terms": int((df > 0).sum()),
                      "terms_with_fewer_than_8_postings": int(((df > 0) & (df < 8)).sum()), "max_df": int(df.max()),
                      "postings_bytes": pb, "term_metadata_bytes": mb, "bracket_table_bytes": tb,
                      "dense_table_bytes_r2_layout": int(df.shape[0]) * ((rows + 2047) // 2048 + 1) * 4}}
# Q = 1024 dense first, then the 256-query blocks (queries 0..255 of the same batch)
full = dict(st)
blocks = {}
from optimized_rag_amd.sharded import ShardedDenseIndex  # noqa: E402
import bench_modes as BM  # noqa: E402
dense = ShardedDenseIndex(eng, rank=0, world=1)
t = BM.timed_all_ranks(lambda: dense.search(st["queries"], 20), 3, 1, 1)
blocks["dense_q1024"] = {"queries_per_sec": round(1024 / t, 1), "ms_per_batch": round(t * 1e3, 3),
                         "mfma_roof_frac": round(2.0 * 1024 * rows * BE.DIM / t / 1e12 / BE.PEAK_MFMA_TFLOPS, 4)}
st256 = dict(st, queries=st["queries"][:256].contiguous(), planted=st["planted"][:256], ptr_d=st["ptr_d"][:257].contiguous(),
             q_tok_d=st["q_tok_d"][:256].contiguous(), q_len_d=st["q_len_d"][:256].contiguous())
blocks.update(BS.shard_blocks(eng, st256, dev, steps=3))
out["one_gpu_share"] = blocks
log("share measured", blocks)
# ---- the token store as an 8-GPU node replicates it: every rank holds ALL passages ------------------------------------------
q, pool, k, L = st256["queries"], 100, 20, 256
ids1, sc1, lg1, cand1 = [x.clone() for x in eng.retrieve_rerank_dev(q, st256["q_tok_d"], st256["q_len_d"], pool, k, term_ptr=st256["ptr_d"],
                                                                    terms=st["terms_d"], L_pair=L)]
total = rows * replicate_world
t0 = time.perf_counter()
eng.tokens_reserve(total, BS.TOK_L)
for c in range((total + BS.TOK_CHUNK - 1) // BS.TOK_CHUNK):
    tok, ln = BS.gen_tokens_chunk(c, min(BS.TOK_CHUNK, total - c * BS.TOK_CHUNK), dev, st["cfg"]["vocab_size"])
    eng.tokens_append_dev(tok, ln)
del tok, ln
torch.cuda.empty_cache()
rep_s = time.perf_counter() - t0
pipe = ShardedPipeline(eng, rank=0, world=1)
run = lambda: pipe.retrieve_rerank(q, st256["ptr_d"], st["terms_d"], st256["q_tok_d"], st256["q_len_d"], pool, k, L_pair=L)
t = BM.timed_all_ranks(run, 2, 1, 1)
ids2, sc2, lg2, cand2 = run()
torch.cuda.synchronize()
used, cap = BS.hbm_used_gb(dev)
out["replicated_token_store"] = {"passages": total, "token_store_bytes": total * BS.TOK_L * 2, "load_s": round(rep_s, 1),
                                 "retrieve_rerank_q256": {"queries_per_sec": round(256 / t, 2), "ms_per_batch": round(t * 1e3, 2)},
                                 "equals_one_call_entry": bool(torch.equal(ids1, ids2) and torch.equal(cand1, cand2) and torch.equal(lg1, lg2)),
                                 "hbm_used_gb": used, "hbm_total_gb": cap}
print(json.dumps(out))
