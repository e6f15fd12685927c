"""CPU tests of the bulk-load shard format: export of document_chunks-shaped rows, pgvector text
literals, memory-mapped read side, temporal vector, doc-partitioned BM25 slices."""
import os
from datetime import datetime, timedelta

import numpy as np
import pytest

from optimized_rag_amd import shard_format as SF
from optimized_rag_amd.bm25 import Bm25Postings
from oracle import rag_oracle as O


def make_rows(rng, n, dim):
    words = [f"w{i}" for i in range(30)]
    now = datetime(2026, 1, 1, 12, 0, 0)
    rows, emb = [], rng.standard_normal((n, dim)).astype(np.float32)
    for i in range(n):
        created = None if i % 7 == 3 else now - timedelta(days=float(rng.uniform(0, 90)))
        rows.append((1000 + 3 * i, f"agent-{i % 3}", " ".join(rng.choice(words, size=int(rng.integers(2, 9)))),
                     SF.format_pgvector_text(emb[i]) if i % 2 else emb[i].tolist(),
                     '{"page": %d}' % i if i % 2 else {"page": i}, created))
    return rows, emb, now


def test_pgvector_text_round_trip():
    rng = np.random.default_rng(1)
    v = (rng.standard_normal(1536) * rng.choice([1e-6, 1.0, 1e4], 1536)).astype(np.float32)
    v[:3] = [0.0, -1.0, 3.0]
    np.testing.assert_array_equal(SF.parse_pgvector_text(SF.format_pgvector_text(v), 1536), v)
    np.testing.assert_array_equal(SF.parse_pgvector_text(" [1,2.5,-3e-2] "), np.array([1, 2.5, -0.03], np.float32))
    with pytest.raises(ValueError):
        SF.parse_pgvector_text("[1,2]", dim=3)
    with pytest.raises(ValueError):
        SF.parse_pgvector_text("1,2,3")


def test_export_open_round_trip(tmp_path):
    rng = np.random.default_rng(2)
    rows, emb, now = make_rows(rng, 57, 48)
    path = SF.export_table(rows, str(tmp_path / "shard"), dim=48)
    sh = SF.open_shard(path)
    assert (sh.n_rows, sh.dim) == (57, 48)
    np.testing.assert_array_equal(np.asarray(sh.embeddings), emb)                       # both input forms, bit for bit
    np.testing.assert_array_equal(np.asarray(sh.ids), [1000 + 3 * i for i in range(57)])
    assert sh.tenant_table == {"agent-0": 0, "agent-1": 1, "agent-2": 2}
    np.testing.assert_array_equal(np.asarray(sh.tenants), [i % 3 for i in range(57)])
    for i in (0, 1, 20, 56):
        r = sh.row(i)
        assert r["content"] == rows[i][2] and r["metadata"] == {"page": i}
    assert np.isnan(sh.created_at[3]) and not np.isnan(sh.created_at[4])
    # temporal vector == the oracle's per-document formula (retrieval.py:266-292)
    md = [{"created_at": r[5].isoformat()} if r[5] is not None else {} for r in rows]
    want = O.temporal_scores(57, md, now)
    np.testing.assert_allclose(sh.temporal_scores(now), want, rtol=0, atol=1e-12)   # epoch-second arithmetic: ~1e-14
    # BM25 CSR survives the trip
    p, q = sh.postings(), Bm25Postings.from_corpus([r[2] for r in rows])
    for a, b in ((p.indptr, q.indptr), (p.doc, q.doc), (p.tf, q.tf), (p.doc_len, q.doc_len), (p.idf, q.idf)):
        np.testing.assert_array_equal(a, b)
    assert p.avgdl == q.avgdl and p.vocab == q.vocab
    sh.close()
    assert sorted(os.listdir(path)) == ["bm25.npz", "created_at.npy", "embeddings.npy", "ids.npy", "meta.json",
                                        "rows.idx.npy", "rows.jsonl", "tenants.npy"]


def test_postings_shard_keeps_global_statistics():
    rng = np.random.default_rng(3)
    words = [f"w{i}" for i in range(25)]
    corpus = [" ".join(rng.choice(words, size=int(rng.integers(1, 10)))) for _ in range(90)]
    post = Bm25Postings.from_corpus(corpus)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    query = "w3 w7 w7 w11 zzz"
    full = obm.get_scores(O.tokenize(query))
    ptr, terms = post.encode_queries([query])
    for b, e in ((0, 31), (31, 60), (60, 90)):
        s = post.shard(b, e)
        assert s.n_docs == e - b and s.idf is post.idf and s.avgdl == post.avgdl
        sc = np.zeros(e - b)
        for t in terms:                                       # raw BM25 of the slice with the global idf / avgdl
            if t < 0:
                continue
            lo, hi = int(s.indptr[t]), int(s.indptr[t + 1])
            d, f = s.doc[lo:hi], s.tf[lo:hi].astype(np.float64)
            sc[d] += s.idf[t] * (f * (s.k1 + 1) / (f + s.k1 * (1 - s.b + s.b * s.doc_len[d] / s.avgdl)))
        np.testing.assert_array_equal(sc, full[b:e])          # bit-identical to the unsharded scores of those docs
