"""The per-GPU share of BASELINE.json configs[4] end to end on ONE GPU: dense index + doc-partitioned BM25
postings over a 2M-term Zipf vocabulary + passage token store + cross-encoder, searched through rag_hybrid_rrf_dev and
rag_retrieve_rerank_dev at Q = 256, pool = 100, against the whole oracle composition on sampled queries: float64 dense
top-100 (exact rescoring of a float32 shortlist), CSR BM25Okapi top-100, RRF ranks (candidate list BIT-EXACT),
'longest_first' pair assembly, float64 BERT forward, sigmoid, stable sort (scores <= 1e-3, logits <= 4e-3).
Reference path: /root/reference/rag/document_store.py:448-460, rag/retrieval.py:324-347, rag/reranker.py:224-271,346-359.

Since round 4 the default suite runs it at the REAL size (12,500,000 rows = 115 GB of embeddings, ~1.07e9 postings, the
share's own 12.5M passages: ~160 GB of HBM, about half a minute on an MI355X); RAG_TEST_SHARD_ROWS=250000 is the quick form of the same
code path for development boxes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS = int(os.environ.get("RAG_TEST_SHARD_ROWS", "12500000"))


def test_per_gpu_share_hybrid_and_rerank_vs_oracle_composition():
    import torch
    import bench as BE
    import bench_shard as BS
    from optimized_rag_amd import RagEngine
    from optimized_rag_amd._lib import bm25_index_bytes
    from oracle import bert_oracle as B
    from oracle import rag_oracle as O
    dev = torch.device("cuda", 0)
    Q, pool, k, L, Lq = 256, 100, 20, 256, 16
    eng = RagEngine(dim=BE.DIM, device=0)
    try:
        st = BS.build_shard(eng, dev, ROWS, Q=Q, log=print)
        post = st["post"]
        pb, mb, tb = bm25_index_bytes(post.indptr, ROWS)
        assert tb <= post.indptr[-1], (pb, mb, tb)                 # bracket tables <= 1/12 of the postings at any vocabulary
        keys, rrf, ranks = eng.hybrid_rrf_dev(st["queries"], st["ptr_d"], st["terms_d"], pool, k)
        torch.cuda.synchronize()
        keys_h = keys.cpu().numpy().copy()
        dstats = eng.dense_stats()
        assert dstats["exact_scan"] == 0, dstats
        ids, sc, lg, cand = eng.retrieve_rerank_dev(st["queries"], st["q_tok_d"], st["q_len_d"], pool, k, term_ptr=st["ptr_d"],
                                                    terms=st["terms_d"], L_pair=L)
        torch.cuda.synchronize()
        ids, sc, lg, cand = ids.cpu().numpy(), sc.cpu().numpy(), lg.cpu().numpy(), cand.cpu().numpy()
        assert (ids >= 0).all() and (cand >= 0).all() and (np.diff(sc, axis=1) <= 0).all()
        assert all(set(ids[q]) <= set(cand[q]) and len(set(ids[q])) == k for q in range(Q))
        np.testing.assert_array_equal(keys_h, cand[:, :k])         # the hybrid call's top-20 = the head of the pipeline's candidates
        # ---- oracle composition on sampled queries ------------------------------------------------------------------------
        sel = [3, 200]
        qsel = st["queries"][sel]
        approx = torch.empty((len(sel), ROWS), dtype=torch.float32, device=dev)        # float32 scores of every row (chunks regenerated)
        for c in range(ROWS // BE.CHUNK_ROWS):
            blk = BE.gen_chunk(c, BE.CHUNK_ROWS, dev, "iid", ROWS)
            approx[:, c * BE.CHUNK_ROWS:(c + 1) * BE.CHUNK_ROWS] = qsel @ blk.T
        short = torch.topk(approx, 600, dim=1).indices.cpu().numpy()
        w = None
        for j, qi in enumerate(sel):
            rows_s = np.sort(short[j])
            hq = st["queries"][qi:qi + 1].cpu().numpy()
            exact = O.cosine_matrix(hq, eng.fetch_rows(rows_s))[0]
            assert np.sort(exact)[-pool] - exact.min() > 2e-3      # the 100th best sits far above the shortlist's tail: float32 error << margin
            d_rows = rows_s[np.lexsort((rows_s, -exact))[:pool]]
            qt = st["terms"][st["term_ptr"][qi]:st["term_ptr"][qi + 1]].tolist()
            raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, qt)
            b_rows = O.stable_topk_desc(raw, pool)
            okeys, _, _ = O.rrf_fuse([[int(r) for r in d_rows], [int(r) for r in b_rows]], k=60, top_k=pool)
            assert cand[qi].tolist() == okeys                                             # candidate list: bit-exact
            if w is None:
                cfg = B.minilm_config()
                from optimized_rag_amd.cross_encoder import LAYER_KEYS
                names = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
                         "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias"]
                for l in range(cfg["layers"]):
                    names += [f"bert.encoder.layer.{l}.{kk}" for kk in LAYER_KEYS]
                names += ["bert.pooler.dense.weight", "bert.pooler.dense.bias", "classifier.weight", "classifier.bias"]
                w = dict(zip(names, st["tensors"]))                 # the oracle takes the HF state-dict form of the same tensors
            pid = np.zeros((pool, L), dtype=np.int64)
            ptt = np.zeros((pool, L), dtype=np.int64)
            plen = np.zeros(pool, dtype=np.int64)
            for jj, r in enumerate(okeys):
                tok, ln = BS.gen_tokens_chunk(r // BS.TOK_CHUNK, min(BS.TOK_CHUNK, ROWS - (r // BS.TOK_CHUNK) * BS.TOK_CHUNK), dev, cfg["vocab_size"])
                trow, tlen = tok[r % BS.TOK_CHUNK].cpu().numpy(), int(ln[r % BS.TOK_CHUNK])
                ql, dl = O.longest_first_lengths(Lq, tlen, L - 3)
                row = [101] + st["q_tok"][qi, :ql].tolist() + [102] + trow[:dl].tolist() + [102]
                pid[jj, :len(row)] = row
                ptt[jj, ql + 2:len(row)] = 1
                plen[jj] = len(row)
            ologit = B.forward_logits(w, cfg, pid, ptt, plen, fast_erf=True)
            oscore = np.array([O.sigmoid(float(x)) for x in ologit])
            order = sorted(range(pool), key=lambda jj: -oscore[jj])
            np.testing.assert_allclose(sc[qi], oscore[order[:k]], atol=1e-3)
            np.testing.assert_allclose(lg[qi], ologit[order[:k]], atol=4e-3)
            if np.abs(np.diff(oscore[order[:k + 1]])).min() > 2e-3:
                assert ids[qi].tolist() == [okeys[jj] for jj in order[:k]]
    finally:
        eng.close()
