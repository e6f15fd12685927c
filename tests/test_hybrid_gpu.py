"""GPU parity for fusion (RRF, linear), BM25 and the Python mirror classes: HIP path through the C-ABI vs the
golden vectors generated from the reference and vs the CPU oracle. Integer ranks / id lists bit-exact; float64
scores bit-exact where the operation order is reproduced (RRF, BM25, linear fusion), else within 1e-12."""
import json
import os
from datetime import datetime

import numpy as np
import pytest

from oracle import rag_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from optimized_rag_amd import RagEngine
    e = RagEngine(dim=1536, device=0)
    yield e
    e.close()


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------------- RRF
def test_rrf_golden_bit_exact(eng, golden_dir):
    from optimized_rag_amd.reranker import ReciprocalRankFusion
    g = load(golden_dir, "rrf.json")
    for c in g["cases"]:
        lists = [[{"content": f"doc-{i}", "id": i} for i in l] for l in c["lists"]]
        fused = ReciprocalRankFusion(k=c["k"], engine=eng).fuse(lists, top_k=c["top_k"])
        assert [d["id"] for d in fused] == c["expected_ids"]
        assert [d["rrf_score"] for d in fused] == c["expected_scores"]          # bit-exact float64
    d = g["dup"]
    lists = [[({"content": x} if x is not None else {}) for x in l] for l in d["lists"]]
    fused = ReciprocalRankFusion(k=60, engine=eng).fuse(lists, top_k=10)
    assert [x.get("content", "") for x in fused] == d["expected_contents"]
    assert [x["rrf_score"] for x in fused] == d["expected_scores"]
    assert ReciprocalRankFusion(engine=eng).fuse([[], []]) == []


def test_rrf_batched_ranks_vs_oracle(eng):
    """BASELINE config 3 shape: Q queries x 2 lists (dense, bm25) x K'=100 -> top-20, ranks bit-exact."""
    rng = np.random.default_rng(1)
    Q, L, ln, k = 64, 2, 100, 20
    lists = np.stack([np.stack([rng.permutation(400)[:ln] for _ in range(L)]) for _ in range(Q)]).astype(np.int64)
    lists[3, 1, 60:] = -1                                      # ragged list
    lists[4, 0, :] = -1                                        # empty list
    keys, scores, ranks = eng.rrf_fuse(lists, rrf_k=60, top_k=k)
    for q in range(Q):
        ls = [[int(x) for x in lists[q, l] if x >= 0] for l in range(L)]
        okeys, oscores, oranks = O.rrf_fuse(ls, k=60, top_k=k)
        n = len(okeys)
        assert keys[q, :n].tolist() == okeys and (keys[q, n:] == -1).all()
        assert scores[q, :n].tolist() == oscores
        assert ranks[q, :n].tolist() == oranks


# ---------------------------------------------------------------------------------------- hybrid_search
def _hr(eng, c):
    from optimized_rag_amd.retrieval import HybridRetriever
    a, b, g_ = c["alpha_beta_gamma_default"]
    hr = HybridRetriever(None, None, "agent-x", a, b, g_, use_adaptive_weights=c["use_adaptive_weights"], engine=eng)
    hr._now = lambda: datetime.fromisoformat(c["now"])
    return hr


def test_hybrid_search_golden_keyword_path(eng, golden_dir):
    """The fixture was produced by the reference with rank_bm25 absent (keyword-overlap path)."""
    g = load(golden_dir, "hybrid_search.json")
    for c in g["cases"]:
        hr = _hr(eng, c)
        hr.bm25_available = False
        emb = [[float(np.float32(x)) for x in row] for row in c["embeddings"]]
        out = hr.hybrid_search(c["query"], c["corpus"], emb, [float(np.float32(x)) for x in c["query_embedding"]],
                               top_k=c["top_k"], documents_metadata=c["metadata"], query_intent=c["intent"])
        assert [r["content"] for r in out] == [c["corpus"][i] for i in c["expected_idx"]]
        for r, e, hm in zip(out, c["expected"], c["expected_has_metadata"]):
            for key in e:
                assert abs(r[key] - e[key]) < 1e-12, key
            assert ("metadata" in r) == hm
        # identity of the returned rows (duplicates keep input order)
        assert [r["embedding"] is emb[i] for r, i in zip(out, c["expected_idx"])] == [True] * len(out)
    for kc in g["keyword"]:
        assert _hr(eng, g["cases"][0])._simple_keyword_scores(kc["query"], kc["corpus"]) == kc["expected"]


def test_hybrid_search_bm25_path_vs_oracle(eng):
    import tools_textgen as T
    rng = np.random.default_rng(11)
    corpus = [T.make_doc(rng, int(rng.integers(1, 6))) for _ in range(200)]
    corpus[7] = corpus[3]
    corpus[50] = ""
    emb = rng.standard_normal((200, 256)).astype(np.float32)
    qe = rng.standard_normal(256).astype(np.float32)
    from optimized_rag_amd.retrieval import HybridRetriever
    hr = HybridRetriever(None, None, "a", engine=eng)
    for query, intent in [("memory vector index", "search"), ("paris paris london", None), ("zzz unknown", "summarization")]:
        out = hr.hybrid_search(query, corpus, emb.tolist(), qe.tolist(), top_k=15, query_intent=intent)
        idx, rows = O.hybrid_search(query, corpus, emb, qe, top_k=15, intent=intent, bm25_available=True)
        assert [r["content"] for r in out] == [corpus[i] for i in idx]
        for r, e in zip(out, rows):
            assert r["keyword_score"] == e["keyword_score"]                 # BM25 float64 bit-exact
            assert abs(r["semantic_score"] - e["semantic_score"]) < 1e-12
            assert abs(r["hybrid_score"] - e["hybrid_score"]) < 1e-12
    assert hr._bm25_scores("q", ["", "  "]) == [0.0, 0.0]
    assert hr.hybrid_search("q", [], [], qe.tolist()) == []


# ---------------------------------------------------------------------------------------- BM25 index path
def synthetic_postings(rng, n_docs, vocab, mean_len):
    """doc length ~ Poisson(mean_len), tokens Zipf(1.1) over `vocab` ids; returns docs as token-id lists."""
    lens = rng.poisson(mean_len, n_docs)
    docs = []
    for L in lens:
        t = rng.zipf(1.1, int(L)) - 1
        docs.append([int(x) % vocab for x in t])
    return docs


def test_bm25_topk_vs_oracle(eng):
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(21)
    n_docs = 40000                                               # 20 doc ranges of 2048
    docs = synthetic_postings(rng, n_docs, 5000, 30)
    docs[100] = []                                               # empty doc
    corpus = [" ".join(f"t{t}" for t in d) for d in docs]
    post = Bm25Postings.from_corpus(corpus).load(eng)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    np.testing.assert_array_equal(post.idf, [obm.idf[w] for w in post.vocab])          # idf table bit-exact
    queries = []
    for _ in range(12):
        d = docs[int(rng.integers(0, n_docs))] or [1]
        queries.append(" ".join(f"t{t}" for t in rng.choice(d, size=min(len(d), int(rng.integers(1, 9))))))
    queries += ["t0 t0 t1", "nosuchtoken", "t4999 nosuchtoken t3"]
    ptr, terms = post.encode_queries(queries)
    k = 100
    ids, rows, scores, mx = eng.bm25_topk(ptr, terms, k)
    dense = eng.bm25_scores(ptr, terms)
    for qi, q in enumerate(queries):
        raw = obm.get_scores(O.tokenize(q))
        np.testing.assert_array_equal(dense[qi], raw)                                # float64 bit-exact, all docs
        m = raw.max() if raw.max() > 0 else 1.0
        assert mx[qi] == m
        top = O.stable_topk_desc(raw, k)
        np.testing.assert_array_equal(rows[qi], top.astype(np.int32))
        np.testing.assert_array_equal(scores[qi], raw[top] / m)


def test_bm25_edge_fewer_docs_than_k(eng):
    from optimized_rag_amd.bm25 import Bm25Postings
    corpus = ["a b b c", "a d", "b b b b e f", "g"]
    post = Bm25Postings.from_corpus(corpus).load(eng)
    ptr, terms = post.encode_queries(["b c c zzz", ""])
    ids, rows, scores, mx = eng.bm25_topk(ptr, terms, 6)
    exp = O.bm25_scores("b c c zzz", corpus)                    # idf("b") is exactly 0.0 here: only doc 0 scores
    order = [int(i) for i in O.stable_topk_desc(exp, 6)]
    assert rows[0].tolist() == order + [-1, -1]
    assert scores[0][:4].tolist() == [exp[i] for i in order] and scores[0][0] == 1.0
    assert rows[1].tolist() == [0, 1, 2, 3, -1, -1] and mx[1] == 1.0              # empty query: all zeros, index order


def test_linear_fuse_large_n_ties(eng):
    rng = np.random.default_rng(31)
    n = 50000
    sem = np.round(rng.uniform(-1, 1, n), 2)                      # many exact ties
    kw = np.round(rng.uniform(0, 1, n), 1)
    tmp = rng.uniform(0, 0.15, n) * (rng.uniform(size=n) < 0.1)
    idx, hyb = eng.linear_fuse_topk(sem, kw, tmp, 0.55, 0.35, 0.10, 200)
    exp = [0.55 * sem[i] + 0.35 * kw[i] + 0.10 * tmp[i] for i in range(n)]
    assert hyb.tolist() == exp                                    # same float64 operations as CPython
    assert idx.tolist() == [int(i) for i in O.stable_topk_desc(exp, 200)]


# ---------------------------------------------------------------------------------------- MMR, rerankers
def test_mmr_both_variants_golden(eng, golden_dir):
    from optimized_rag_amd.nodes_helpers import apply_mmr
    from optimized_rag_amd.reranker import MMRDiversifier
    g = load(golden_dir, "mmr.json")
    for c in g["class"]:
        docs = [{"content": f"d{i}", "embedding": [float(np.float32(x)) for x in e], "pos": i} for i, e in enumerate(c["emb"])]
        out = MMRDiversifier(c["lambda"], engine=eng).diversify([float(np.float32(x)) for x in c["q"]], docs, top_k=c["top_k"])
        assert [d["pos"] for d in out] == c["expected_pos"]
        np.testing.assert_allclose([d["mmr_score"] for d in out], c["expected_mmr"], atol=1e-12)
    docs = [{"content": "a", "embedding": [1.0, 0.0]}, {"content": "b", "embedding": []},
            {"content": "c", "embedding": [float("nan"), 1.0]}, {"content": "d"},
            {"content": "e", "embedding": [0.6, 0.8]}, {"content": "f", "embedding": [float("inf"), 1.0]},
            {"content": "g", "embedding": (1.0, 0.0)}]
    out = MMRDiversifier(0.7, engine=eng).diversify([1.0, 0.2], docs, top_k=5)
    assert [d["content"] for d in out] == g["invalid"]["expected_contents"]
    np.testing.assert_allclose([d["mmr_score"] for d in out], g["invalid"]["expected_mmr"], atol=1e-12)
    out2 = MMRDiversifier(0.7, engine=eng).diversify([1.0, 0.2], [{"content": "b", "embedding": []}, {"content": "d"}], top_k=1)
    assert [d["content"] for d in out2] == g["invalid"]["none_valid_contents"]
    for c in g["helper"]:
        q = [float(np.float32(x)) for x in c["q"]]

        class Svc:
            def generate_embedding(self, text):
                return q

        docs = [{"content": f"d{i}", "embedding": [float(np.float32(x)) for x in e], "pos": i} for i, e in enumerate(c["emb"])]
        out = apply_mmr("the query", docs, c["lambda"], c["k"], Svc(), engine=eng)
        assert [d["pos"] for d in out] == c["expected_pos"]


def test_rerankers_golden(eng, golden_dir):
    from optimized_rag_amd.reranker import CrossEncoderReranker, OpenAIReranker
    g = load(golden_dir, "rerankers.json")
    oai = g["openai"]
    emb = np.array(oai["emb"], dtype=np.float32)

    class Item:
        def __init__(self, e):
            self.embedding = e

    class Client:
        class embeddings:
            @staticmethod
            def create(input, model):
                class R:
                    data = [Item([float(x) for x in emb[i]]) for i in range(len(input))]
                return R()

    res = [dict(d) for d in oai["results"]]
    out = OpenAIReranker(Client(), "m", engine=eng).rerank("q", res, top_k=oai["top_k"])
    assert [d["pos"] for d in out] == oai["expected_pos"]
    np.testing.assert_allclose([d["rerank_score"] for d in out], oai["expected_rerank"], atol=1e-12)
    assert all("embedding" in d for d in res)

    class Bad:
        class embeddings:
            @staticmethod
            def create(input, model):
                raise RuntimeError("down")

    assert [d["pos"] for d in OpenAIReranker(Bad(), "m", engine=eng).rerank("q", [dict(d) for d in oai["results"]], top_k=4)] == oai["fail_pos"]

    cr = g["cross"]
    ce = CrossEncoderReranker(model_name="cross-encoder/ms-marco-MiniLM-L-6-v2", engine=eng)     # not a local dir
    assert ce.is_available() is False
    assert [d["pos"] for d in ce.rerank("q", [dict(d) for d in cr["docs"]], top_k=3)] == cr["fallback_pos"]
    seen = {}

    class Fake:
        def predict(self, pairs):
            seen["len"] = [len(p[1]) for p in pairs]
            return np.array(cr["logits"], dtype=np.float32)

    ce.model = Fake()
    out = ce.rerank("the query", [dict(d) for d in cr["docs"]], top_k=cr["top_k"])
    assert seen["len"] == cr["pairs_len"]
    for d, e in zip(out, cr["expected"]):
        for key in ("pos", "score", "cross_encoder_score", "cross_encoder_raw_score"):
            assert d[key] == e[key]
        assert d.get("embedding_score") == e["embedding_score"]


# ---------------------------------------------------------------------------------------- consistency / compressor
def test_consistency_golden(eng, golden_dir):
    from optimized_rag_amd.consistency_checker import ConsistencyChecker
    g = load(golden_dir, "consistency.json")
    for c in g["cases"]:
        table = c["embeddings"]

        class Svc:
            def generate_embeddings_batch(self, texts):
                return [table[t] for t in texts]

        chk = ConsistencyChecker(Svc(), similarity_threshold=c["threshold"], engine=eng)
        for d, exp in zip(c["docs"], c["expected_claims"]):
            assert chk._extract_claims(d["content"]) == exp
        out = chk.check_consistency([dict(d) for d in c["docs"]], "some query")
        exp = c["expected"]
        assert abs(out.pop("confidence") - exp.pop("confidence")) < 1e-12
        assert out == exp
    e = g["edge"]
    chk = ConsistencyChecker(None, engine=eng)
    assert chk.check_consistency([{"content": "x"}], "q") == e["one_doc"]
    assert chk.check_consistency([{"content": "Tiny."}, {"content": "Also tiny."}], "q") == e["few_claims"]

    class Boom:
        def generate_embeddings_batch(self, t):
            raise RuntimeError("embedding backend down")

    assert ConsistencyChecker(Boom(), engine=eng).check_consistency(e["embed_fail_docs"], "q") == e["embed_fail"]
    for p in e["is_contradiction"]:
        assert chk._is_contradiction(p["a"], p["b"]) == p["expected"]


def test_compressor_golden(eng, golden_dir):
    from optimized_rag_amd.context_compressor import ContextCompressor
    g = load(golden_dir, "compressor.json")

    def svc_for(table):
        class Svc:
            def generate_embedding(self, t):
                return table[t]

            def generate_embeddings_batch(self, ts):
                return [table[t] for t in ts]
        return Svc()

    for c in g["cases"]:
        comp = ContextCompressor(max_tokens=c["max_tokens"], sentences_per_doc=c["sentences_per_doc"],
                                 embedding_service=svc_for(c["embeddings"]), conservative_mode=c["conservative"], engine=eng)
        out = comp.compress(c["query"], [dict(d) for d in c["docs"]], query_intent="question_answering", confidence=c["confidence"])
        assert out == c["expected"]
        assert comp.get_compression_stats(out) == c["expected_stats"]
    sh = g["score_hybrid"]
    comp = ContextCompressor(embedding_service=svc_for(sh["embeddings"]), conservative_mode=False, engine=eng)
    got = [s for _, s in comp._score_sentences_hybrid(sh["query"], sh["sentences"])]
    np.testing.assert_allclose(got, sh["expected"], atol=1e-12)
    for c in g["lexical"]:
        assert abs(comp._score_sentence_lexical(c["q"], c["s"]) - c["expected"]) < 1e-15
    for c in g["split"]:
        assert comp._split_sentences(c["text"]) == c["expected"]
    lo = g["lexical_only"]
    nos = ContextCompressor(sentences_per_doc=2, embedding_service=None, conservative_mode=False, engine=eng)
    assert nos.compress(lo["query"], [dict(d) for d in lo["docs"]]) == lo["expected"]


# ---------------------------------------------------------------------------------------- document index
def test_gpu_document_index_search(eng):
    from optimized_rag_amd.document_store import GpuDocumentIndex
    from optimized_rag_amd.retrieval import HybridRetriever
    rng = np.random.default_rng(41)
    N, D = 5000, 1536
    emb = rng.standard_normal((N, D)).astype(np.float32)
    rows = [{"content": f"chunk {i}", "agent_id": f"agent-{i % 3}", "filename": f"f{i % 7}.pdf", "file_type": "pdf",
             "metadata": {"i": i}, "id": 10_000 + i} for i in range(N)]
    q = (emb[1234] + 0.3 * rng.standard_normal(D)).astype(np.float32)

    class Svc:
        def generate_embedding(self, text):
            return [float(x) for x in q]

    store = GpuDocumentIndex(Svc(), dim=D, engine=eng)
    store.bulk_load(rows, emb)
    tenant = np.array([i % 3 for i in range(N)])
    out = store.search("agent-1", "whatever", top_k=5)
    oid, osc = O.dense_topk(emb, q[None], 5, tenant_of_row=tenant, tenant=1)
    assert [d["metadata"]["i"] for d in out] == oid[0].tolist()
    np.testing.assert_allclose([d["score"] for d in out], osc[0], atol=1e-9)
    assert set(out[0]) == {"content", "filename", "file_type", "score", "metadata", "embedding"}
    assert out[0]["embedding"] == [float(x) for x in emb[oid[0][0]]]
    assert store.search("no-such-agent", "q") == []
    arch = store.search_archival_memory("agent-0", [float(x) for x in q], limit=3)
    oid0, _ = O.dense_topk(emb, q[None], 3, tenant_of_row=tenant, tenant=0)
    assert [a["id"] for a in arch] == [10_000 + int(i) for i in oid0[0]]
    hr = HybridRetriever(memory_manager=None, document_store=store, agent_id="agent-1", engine=eng)
    res = hr.retrieve("whatever", sources=["documents"], top_k=5)
    assert [d["metadata"]["i"] for d in res] == oid[0].tolist() and all(d["source"] == "documents" for d in res)


def test_batched_agent_entries_equal_the_per_query_calls(eng):
    """GpuDocumentIndex.search_many / HybridRetriever.retrieve_batch / retrieve_tier_2_batch answer
    a list of queries with ONE dense search and must return, element by element, what the reference-shaped single-query
    calls return (same rows, same float64 scores, same dict keys, `source` / `tier` tags)."""
    from optimized_rag_amd.document_store import GpuDocumentIndex
    from optimized_rag_amd.hierarchical import retrieve_tier_2_batch
    from optimized_rag_amd.retrieval import HybridRetriever
    rng = np.random.default_rng(43)
    N, D, Qn = 6000, 1536, 9
    emb = rng.standard_normal((N, D)).astype(np.float32)
    rows = [{"content": f"chunk {i}", "agent_id": f"agent-{i % 2}", "filename": "f.txt", "file_type": "txt", "metadata": {"i": i}}
            for i in range(N)]
    table = {f"question {j}": [float(x) for x in (emb[100 * j + 3] + 0.4 * rng.standard_normal(D)).astype(np.float32)] for j in range(Qn)}
    calls = {"single": 0, "batch": 0}

    class Svc:
        def generate_embedding(self, text):
            calls["single"] += 1
            return table[text]

        def generate_embeddings_batch(self, texts):
            calls["batch"] += 1
            return [table[t] for t in texts]

    store = GpuDocumentIndex(Svc(), dim=D, engine=eng)
    store.bulk_load(rows, emb)
    qs = list(table)
    many = store.search_many("agent-1", qs, top_k=7)
    assert calls == {"single": 0, "batch": 1}
    singles = [store.search("agent-1", q, top_k=7) for q in qs]
    assert many == singles
    hr = HybridRetriever(memory_manager=None, document_store=store, agent_id="agent-1", engine=eng)
    assert hr.retrieve_batch(qs, sources=["documents"], top_k=7) == [hr.retrieve(q, sources=["documents"], top_k=7) for q in qs]
    tiered = retrieve_tier_2_batch(hr, qs, 7)
    assert [[d["metadata"]["i"] for d in res] for res in tiered] == [[d["metadata"]["i"] for d in res] for res in singles]
    assert all(d["tier"] == 2 and d["source"] == "documents" for res in tiered for d in res)
    assert store.search_many("no-such-agent", qs[:2]) == [[], []] and store.search_many("agent-1", []) == []
    eng.set_tenants(None)


# ---------------------------------------------------------------------------------------- fused device hybrid
def test_hybrid_rrf_dev_matches_oracle_pipeline(eng, monkeypatch):
    """rag_hybrid_rrf_dev (dense top-pool + BM25 top-pool + RRF, all on device) == oracle dense + oracle BM25 + oracle RRF.
    12 queries <= RAG_FORK_MAX_Q: the BM25 leg runs on the side stream beside the dense leg; option no_fork (both legs in line
    on the caller's stream) must give the same bits."""
    import torch
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(51)
    N, D, Q, pool, k = 20000, 1536, 12, 100, 20
    docs = synthetic_postings(rng, N, 3000, 25)
    corpus = [" ".join(f"t{t}" for t in d) for d in docs]
    emb = rng.standard_normal((N, D)).astype(np.float32)
    q = (emb[rng.integers(0, N, Q)] + 0.5 * rng.standard_normal((Q, D))).astype(np.float32)
    queries = [" ".join(f"t{t}" for t in rng.choice(docs[int(rng.integers(0, N))] or [1], size=5)) for _ in range(Q)]
    ids = (np.arange(N)[::-1] + 7_000_000).astype(np.int64)                      # non-trivial id mapping
    eng.index_load(emb, ids=ids)
    post = Bm25Postings.from_corpus(corpus).load(eng)
    ptr, terms = post.encode_queries(queries)
    keys, rrf, ranks = eng.hybrid_rrf_dev(torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(),
                                          torch.from_numpy(terms).cuda(), pool, k)
    torch.cuda.synchronize()
    keys, rrf, ranks = keys.cpu().numpy(), rrf.cpu().numpy(), ranks.cpu().numpy()
    eng.set_option("no_fork", 1)
    try:
        k2, r2, n2 = eng.hybrid_rrf_dev(torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), pool, k)
        torch.cuda.synchronize()
    finally:
        eng.set_option("no_fork", 0)
    np.testing.assert_array_equal(keys, k2.cpu().numpy())
    np.testing.assert_array_equal(rrf, r2.cpu().numpy())
    np.testing.assert_array_equal(ranks, n2.cpu().numpy())
    d_rows, _ = O.dense_topk(emb, q, pool)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    for qi in range(Q):
        b_rows = O.stable_topk_desc(obm.get_scores(O.tokenize(queries[qi])), pool)
        okeys, oscores, oranks = O.rrf_fuse([[int(ids[r]) for r in d_rows[qi]], [int(ids[r]) for r in b_rows]], k=60, top_k=k)
        assert keys[qi].tolist() == okeys
        assert rrf[qi].tolist() == oscores                                    # bit-exact float64
        assert ranks[qi].tolist() == oranks


def test_row_sharded_hybrid_equals_unsharded(eng):
    """SURVEY §8e on one GPU: three engines hold three row shards (embeddings + doc-partitioned postings with the GLOBAL
    idf / avgdl), their local_lists() are stacked as the all-gather would, and fuse_gathered() must reproduce the
    unsharded rag_hybrid_rrf_dev result bit for bit (keys, ranks, float64 RRF scores) and the globally normalised BM25."""
    import torch
    from optimized_rag_amd import RagEngine
    from optimized_rag_amd.bm25 import Bm25Postings
    from optimized_rag_amd.sharded import ShardedHybridIndex, shard_bounds
    rng = np.random.default_rng(77)
    N, D, Q, pool, k, W = 5001, 1536, 14, 50, 20, 3
    docs = synthetic_postings(rng, N, 800, 12)
    corpus = [" ".join(f"t{t}" for t in d) for d in docs]
    emb = rng.standard_normal((N, D)).astype(np.float32)
    emb[4000] = emb[100]                                                          # cross-shard exact dense tie
    q = (emb[rng.integers(0, N, Q)] + 0.5 * rng.standard_normal((Q, D))).astype(np.float32)
    q[0] = emb[100]
    queries = [" ".join(f"t{t}" for t in rng.choice(docs[int(rng.integers(0, N))] or [1], size=4)) for _ in range(Q)]
    queries[1] = "zzz-unknown-token"                                              # all-zero BM25 list: divisor stays 1.0
    post = Bm25Postings.from_corpus(corpus)
    ptr, terms = post.encode_queries(queries)
    qd, pd, td = torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda()

    eng.index_load(emb)
    post.load(eng)
    keys, rrf, ranks = [t.cpu().numpy().copy() for t in eng.hybrid_rrf_dev(qd, pd, td, pool, k)]
    _, _, bsc, _ = eng.bm25_topk(ptr, terms, pool)

    shards, sends = [], []
    for r, (b, e) in enumerate(shard_bounds(N, W)):
        se = RagEngine(dim=D, device=0)
        sh = ShardedHybridIndex(se, rank=r, world=W)
        sh.load_shard(emb[b:e], b, post.shard(b, e))
        sends.append(sh.local_lists(qd, pd, td, pool, k).clone())
        shards.append((se, sh))
    out = shards[0][1].fuse_gathered(torch.stack(sends), k)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out["keys"].cpu().numpy(), keys)
    np.testing.assert_array_equal(out["ranks"].cpu().numpy(), ranks)
    np.testing.assert_array_equal(out["rrf"].cpu().numpy(), rrf)                  # bit-exact float64
    np.testing.assert_array_equal(out["bm25_scores"].cpu().numpy(), bsc)          # same raw / same global max -> same quotient
    for se, _ in shards:
        se.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_mmr_device_selection_matches_oracle(eng, variant):
    """rag_mmr_select_host on explicit candidates and rag_mmr_select_dev on index rows == the oracle's greedy loops
    (positions identical; the scores each pick won with within 1e-12), incl. duplicates, a zero row and -1 padding."""
    import torch
    rng = np.random.default_rng(90 + variant)
    N, D, Q, pool, k, lam = 4000, 1536, 6, 100, 20, 0.7
    centers = rng.standard_normal((12, D))
    emb = (centers[rng.integers(0, 12, N)] + 0.6 * rng.standard_normal((N, D))).astype(np.float32)   # clustered: diversity matters
    emb[17] = emb[5]                                                        # exact duplicate rows
    emb[23] = 0.0                                                           # zero-norm row: cosine 0.0
    q = (centers[:Q] + 0.5 * rng.standard_normal((Q, D))).astype(np.float32)
    eng.index_load(emb)
    rows = np.stack([rng.choice(N, pool, replace=False) for _ in range(Q)]).astype(np.int32)
    rows[0, :4] = [5, 17, 23, 40]
    rows[1, 60:] = -1                                                       # short pool
    rows[2, 3:] = -1                                                        # fewer candidates than k
    sel = torch.empty((Q, k), dtype=torch.int32, device="cuda")
    sc = torch.empty((Q, k), dtype=torch.float64, device="cuda")
    eng.mmr_select_dev(torch.from_numpy(q).cuda(), torch.from_numpy(rows).cuda(), k, lam, variant, sel, sc)
    torch.cuda.synchronize()
    sel, sc = sel.cpu().numpy(), sc.cpu().numpy()
    for qi in range(Q):
        live = [j for j in range(pool) if rows[qi, j] >= 0]
        cand = [emb[rows[qi, j]].astype(np.float64).tolist() for j in live]
        if variant == 0:
            pos, osc = O.mmr_class(q[qi].astype(np.float64).tolist(), cand, k, lam)
        else:
            pos = O.mmr_helper(q[qi].astype(np.float64).tolist(), cand, k, lam) if len(cand) > k else None
            osc = None
        if pos is None:                                   # apply_mmr returns its input unchanged when len <= k (host-side rule)
            continue
        want = [live[p] for p in pos] + [-1] * (k - len(pos))
        assert sel[qi].tolist() == want
        if osc is not None:
            np.testing.assert_allclose(sc[qi, :len(osc)], osc, atol=1e-12)
        # explicit-candidate host entry gives the same picks
        hs, hsc = eng.mmr_select(q[qi], np.asarray([emb[rows[qi, j]] for j in live]), k, lam, variant)
        assert hs.tolist() == pos


@pytest.mark.parametrize("k", [20, 100])
def test_bm25_staged_threshold_path_vs_oracle(eng, k):
    """More than 8 doc ranges (> 131072 docs): the first 4 ranges get the exact per-range select, the others only compact
    keys >= the k-th best of stage one. Results must stay bit-identical to the oracle — including a query whose matches
    are fewer than k (threshold 0: every later range falls back to the exact select), an unknown token (all-zero scores:
    the top-k is the first k docs) and a score plateau far longer than k that straddles many ranges."""
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(31 + k)
    n_docs = 180_000                                             # 88 doc ranges of 2048
    lens = rng.poisson(6, n_docs)
    toks = (rng.zipf(1.3, int(lens.sum())) - 1) % 3000
    ptr_d = np.concatenate([[0], np.cumsum(lens)])
    corpus = [" ".join(f"t{t}" for t in toks[ptr_d[i]:ptr_d[i + 1]]) for i in range(n_docs)]
    for i in range(5000, n_docs, 37):
        corpus[i] = "plateau same words"                          # ~4700 identical docs across all ranges: ties >> k
    corpus[170_000] = "rareword appears once"
    post = Bm25Postings.from_corpus(corpus).load(eng)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    queries = ["t0 t5 t17", "t1 t2 t2 t300", "rareword", "plateau words", "nosuchtoken", "t2999 t40 rareword"]
    for _ in range(6):
        i = int(rng.integers(0, n_docs))
        queries.append(" ".join(rng.choice(corpus[i].split() or ["t1"], size=3)))
    ptr, terms = post.encode_queries(queries)
    ids, rows, scores, mx = eng.bm25_topk(ptr, terms, k)
    for qi, q in enumerate(queries):
        raw = obm.get_scores(O.tokenize(q))
        m = raw.max() if raw.max() > 0 else 1.0
        top = O.stable_topk_desc(raw, k)
        np.testing.assert_array_equal(rows[qi], top.astype(np.int32), err_msg=q)
        np.testing.assert_array_equal(scores[qi], raw[top] / m, err_msg=q)
        assert mx[qi] == m


def test_bm25_bounded_plan_and_sub_batched_device_call(eng):
    """the per-call plan takes plan_t <= 64 token slots per query sized by a byte budget (tokens past it are searched inside
    the scoring kernel) and a device-pointer call whose workspace would exceed its budget runs in query sub-batches. Both are
    forced here on a small index (8 slots against queries of up to 30 tokens; a 1 MiB budget = a handful of queries per sub-batch):
    rows, scores and maxima must be bit-identical to the default call and to the oracle."""
    import torch
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(404)
    n_docs = 60_000                                               # 30 doc ranges: staged path
    lens = rng.poisson(7, n_docs)
    toks = (rng.zipf(1.25, int(lens.sum())) - 1) % 2500
    ptr_d = np.concatenate([[0], np.cumsum(lens)])
    corpus = [" ".join(f"t{t}" for t in toks[ptr_d[i]:ptr_d[i + 1]]) for i in range(n_docs)]
    post = Bm25Postings.from_corpus(corpus).load(eng)
    queries = []
    for n_tok in (1, 3, 8, 9, 17, 30, 5, 12, 30, 2, 64, 70):
        queries.append(" ".join(f"t{t}" for t in (rng.zipf(1.25, n_tok) - 1) % 2500))
    queries = queries * 3                                         # 36 queries
    ptr, terms = post.encode_queries(queries)
    k = 10
    ref = eng.bm25_topk(ptr, terms, k)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    for qi in (0, 5, 10, 11):
        raw = obm.get_scores(O.tokenize(queries[qi]))
        np.testing.assert_array_equal(ref[1][qi], O.stable_topk_desc(raw, k).astype(np.int32))
    ptr_t, terms_t = torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda()

    def dev_call():
        ids = torch.empty((len(queries), k), dtype=torch.int64, device="cuda")
        rows = torch.empty((len(queries), k), dtype=torch.int32, device="cuda")
        sc = torch.empty((len(queries), k), dtype=torch.float64, device="cuda")
        eng.bm25_topk_dev(ptr_t, terms_t, k, ids, rows, sc)
        torch.cuda.synchronize()
        return rows.cpu().numpy(), sc.cpu().numpy()

    base = dev_call()
    np.testing.assert_array_equal(base[0], ref[1])
    np.testing.assert_array_equal(base[1], ref[2])
    for opt, val in (("bm25_plan_slots", 8), ("bm25_ws_mb", 1)):
        eng.set_option(opt, val)
        try:
            got = dev_call()
            host = eng.bm25_topk(ptr, terms, k)
        finally:
            eng.set_option(opt, 0)
        np.testing.assert_array_equal(got[0], ref[1], err_msg=opt)
        np.testing.assert_array_equal(got[1], ref[2], err_msg=opt)
        np.testing.assert_array_equal(host[1], ref[1], err_msg=opt)
        np.testing.assert_array_equal(host[2], ref[2], err_msg=opt)


def _sparse_postings(rng, n_docs, n_terms, per_term):
    """Term-major CSR with a few thousand postings per term spread over ALL doc ranges (cheap to build at millions of docs)."""
    from optimized_rag_amd.bm25 import Bm25Postings
    indptr, docs, tfs = [0], [], []
    for t in range(n_terms):
        d = np.unique(rng.integers(0, n_docs, per_term if t else 40 * per_term))       # term 0 is frequent
        docs.append(d.astype(np.int32))
        tfs.append(rng.integers(1, 4, d.shape[0]).astype(np.int32))
        indptr.append(indptr[-1] + d.shape[0])
    doc_len = rng.integers(5, 40, n_docs).astype(np.int32)
    indptr = np.asarray(indptr, dtype=np.int64)
    idf = Bm25Postings.idf_table(np.diff(indptr), n_docs)
    return Bm25Postings(indptr, np.concatenate(docs), np.concatenate(tfs), doc_len, idf, float(doc_len.sum()) / n_docs)


def test_bm25_more_than_256_ranges_reuses_workspace(eng):
    """above 256 doc ranges (> 524,288 docs at 2048 per range; a 12.5M-row shard has 6104) the merge must still read only the
    COUNTED front of every partial list. The same device workspace is used by two batches with different queries (so
    stale slots of the first batch sit behind the second batch's fronts); both the host and the device entry must equal
    the exact per-range select (option bm25_no_staging) and the CSR oracle, bit for bit."""
    import torch
    rng = np.random.default_rng(404)
    n_docs, k = 4_500_000, 100                                   # 2198 doc ranges (nine merge groups of 256)
    post = _sparse_postings(rng, n_docs, 48, 3000).load(eng)
    batches = [[[0, 3, 7], [5], [1, 1, 2, 40], [47, 0], [-1], [9, 10, 11, 12, 13, 14]],
               [[2, 0], [6, 7, 8], [30, 31], [0], [44, 45, 46, 47], [20]]]
    for terms_of in batches:
        ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
        terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
        ids, rows, scores, mx = eng.bm25_topk(ptr, terms, k)
        Q = len(terms_of)
        di = torch.empty((Q, k), dtype=torch.int64, device="cuda")
        dr = torch.empty((Q, k), dtype=torch.int32, device="cuda")
        ds = torch.empty((Q, k), dtype=torch.float64, device="cuda")
        eng.bm25_topk_dev(torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), k, di, dr, ds)
        torch.cuda.synchronize()
        eng.set_option("bm25_no_staging", 1)
        try:
            _, rows_x, scores_x, mx_x = eng.bm25_topk(ptr, terms, k)
        finally:
            eng.set_option("bm25_no_staging", 0)
        np.testing.assert_array_equal(rows, rows_x)
        np.testing.assert_array_equal(scores, scores_x)
        np.testing.assert_array_equal(dr.cpu().numpy(), rows_x)
        np.testing.assert_array_equal(ds.cpu().numpy(), scores_x)
        for qi, t in enumerate(terms_of):
            raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, t)
            m = raw.max() if raw.max() > 0 else 1.0
            top = O.stable_topk_desc(raw, k)
            np.testing.assert_array_equal(rows[qi], top.astype(np.int32))
            np.testing.assert_array_equal(scores[qi], raw[top] / m)
            assert mx[qi] == m == mx_x[qi]


def test_bm25_and_hybrid_tenant_filter(eng):
    """the `WHERE agent_id = %s` of every reference query (rag/document_store.py:457) must hold for the BM25
    leg too. Tenants: three interleaved ones plus one stored contiguously at the END of the table (the usual export
    order). BM25 top-k under a tenant == oracle scores masked to the tenant's docs (global idf / avgdl), stable order,
    divided by the tenant's own max; the hybrid call returns only that tenant's ids and equals the oracle composition."""
    import torch
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(505)
    N, D, Q, pool, k = 40_000, 1536, 10, 60, 20
    docs = synthetic_postings(rng, N, 3000, 14)
    corpus = [" ".join(f"t{t}" for t in d) for d in docs]
    emb = rng.standard_normal((N, D)).astype(np.float32)
    tenants = rng.integers(0, 3, N).astype(np.int32)
    tenants[N - 700:] = 3                                                          # contiguous tenant, last doc range only
    q = (emb[rng.integers(0, N, Q)] + 0.5 * rng.standard_normal((Q, D))).astype(np.float32)
    queries = [" ".join(f"t{t}" for t in rng.choice(docs[int(rng.integers(0, N))] or [1], size=4)) for _ in range(Q)]
    queries[2] = "nosuchtoken"
    eng.index_load(emb, id_base=1000)
    eng.set_tenants(tenants)
    post = Bm25Postings.from_corpus(corpus).load(eng)
    ptr, terms = post.encode_queries(queries)
    obm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    qd, pd, td = torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda()
    for tenant in (1, 3):
        mine = np.nonzero(tenants == tenant)[0]
        ids, rows, scores, mx = eng.bm25_topk(ptr, terms, pool, tenant=tenant)
        keys, rrf, ranks = [t.cpu().numpy().copy() for t in eng.hybrid_rrf_dev(qd, pd, td, pool, k, tenant=tenant)]
        d_ids, _ = O.dense_topk(emb, q, pool, tenant_of_row=tenants, tenant=tenant)
        for qi in range(Q):
            raw = obm.get_scores(O.tokenize(queries[qi]))[mine]
            m = raw.max() if raw.max() > 0 else 1.0
            top = O.stable_topk_desc(raw, pool)
            np.testing.assert_array_equal(rows[qi], mine[top].astype(np.int32))
            np.testing.assert_array_equal(scores[qi], raw[top] / m)
            assert mx[qi] == m
            okeys, oscores, oranks = O.rrf_fuse([[int(r) + 1000 for r in d_ids[qi] if r >= 0], [int(r) + 1000 for r in mine[top]]],
                                                k=60, top_k=k)
            assert keys[qi].tolist() == okeys and rrf[qi].tolist() == oscores and ranks[qi].tolist() == oranks
            assert all(tenants[key - 1000] == tenant for key in keys[qi] if key >= 0)
    eng.set_tenants(None)


def test_adhoc_hybrid_search_keeps_resident_postings(eng):
    """HybridRetriever.hybrid_search scores its ad-hoc corpus statelessly; the index's resident postings (and
    their normalise flag) survive, and mis-aligned postings are refused by the hybrid call instead of read out of range."""
    import torch
    from optimized_rag_amd import RagError
    from optimized_rag_amd.bm25 import Bm25Postings
    from optimized_rag_amd.retrieval import HybridRetriever
    rng = np.random.default_rng(606)
    N, D = 3000, 1536
    docs = synthetic_postings(rng, N, 500, 10)
    corpus = [" ".join(f"t{t}" for t in d) for d in docs]
    emb = rng.standard_normal((N, D)).astype(np.float32)
    eng.index_load(emb)
    post = Bm25Postings.from_corpus(corpus).load(eng)
    ptr, terms = post.encode_queries(["t1 t2 t3", "t7"])
    before = eng.bm25_topk(ptr, terms, 10)
    hr = HybridRetriever(None, None, "a", engine=eng)
    small = ["alpha beta", "beta gamma gamma", "delta"]
    out = hr.hybrid_search("beta gamma", small, rng.standard_normal((3, 8)).tolist(), rng.standard_normal(8).tolist(), top_k=3)
    assert sorted(r["content"] for r in out) == sorted(small)
    after = eng.bm25_topk(ptr, terms, 10)
    for a, b in zip(before, after):
        np.testing.assert_array_equal(a, b)
    Bm25Postings.from_corpus(small).load(eng)                                     # 3 docs vs 3000 rows: not row-aligned
    qd = torch.from_numpy(emb[:2].copy()).cuda()
    with pytest.raises(RagError, match="row-aligned"):
        eng.hybrid_rrf_dev(qd, torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), 10, 5)


@pytest.mark.parametrize("intent", ["search", "summarization", None])
def test_index_level_linear_hybrid_equals_hybrid_search(eng, intent):
    """rag_hybrid_linear_dev over the RESIDENT index == HybridRetriever.hybrid_search with the whole index
    as its corpus (oracle restatement of rag/retrieval.py:214-322, BM25 path): same rows in the same (stable) order, hybrid
    / keyword / temporal scores bit-identical float64 (same operation order), cosine within 1e-12. N = 4096 rows incl.
    duplicated rows (exact ties -> lower row first), an empty document, a zero embedding, timestamps old / recent / absent."""
    import torch
    from datetime import timedelta
    import tools_textgen as T
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(4096)
    N, D, k = 4096, 1536, 25
    corpus = [T.make_doc(rng, int(rng.integers(1, 4))) for _ in range(N)]
    corpus[77] = ""
    emb = rng.standard_normal((N, D)).astype(np.float32)
    corpus[300], emb[300] = corpus[200], emb[200]                                   # exact duplicate: tie on every component
    emb[500] = 0.0
    now = datetime(2026, 3, 27, 12, 0, 0)
    meta = []
    for i in range(N):
        if i % 5 == 0:
            meta.append({})
        elif i % 5 == 1:
            meta.append({"created_at": (now - timedelta(days=float(rng.uniform(0, 3)))).isoformat()})
        else:
            meta.append({"uploaded_at": (now - timedelta(days=float(rng.uniform(3, 400)))).isoformat()})
    meta[300] = meta[200]
    temporal = np.asarray(O.temporal_scores(N, meta, now), dtype=np.float64)
    queries = ["memory vector index", "paris london paris berlin", "zzz-unknown-token", "kernel bandwidth matrix tile stream"]
    q_emb = (emb[[200, 9, 1000, 4000]] + 0.4 * rng.standard_normal((4, D))).astype(np.float32)
    a, b, g = O.weights_for_intent(intent) if intent else (0.55, 0.35, 0.10)
    eng.index_load(emb)
    eng.set_temporal(temporal)
    post = Bm25Postings.from_corpus(corpus).load(eng)
    ptr, terms = post.encode_queries(queries)
    out = eng.hybrid_linear_dev(torch.from_numpy(q_emb).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), k, a, b, g)
    torch.cuda.synchronize()
    got = {key: v.cpu().numpy() for key, v in out.items()}
    for qi, query in enumerate(queries):
        idx, rows = O.hybrid_search(query, corpus, emb, q_emb[qi], top_k=k, metadata=meta, intent=intent, bm25_available=True, now=now)
        assert got["rows"][qi].tolist() == idx, query
        assert got["ids"][qi].tolist() == idx
        np.testing.assert_allclose(got["semantic"][qi], [r["semantic_score"] for r in rows], atol=1e-12)
        assert got["keyword"][qi].tolist() == [r["keyword_score"] for r in rows]
        assert got["temporal"][qi].tolist() == [r["temporal_score"] for r in rows]
        np.testing.assert_allclose(got["hybrid"][qi], [r["hybrid_score"] for r in rows], atol=1e-12)
    eng.set_temporal(None)


def test_index_level_linear_hybrid_tenant_filter(eng):
    """under a tenant filter the keyword score of rag_hybrid_linear_dev is raw / max over the TENANT's documents
    (the corpus hybrid_search would have been handed behind `WHERE agent_id = %s`, rag/document_store.py:457), as
    rag_bm25_topk_* normalise; idf / avgdl stay the loaded corpus's. Oracle: the CSR BM25 scores, float64 cosines and the
    reference's operation order (rag/retrieval.py:302), stable sort over the tenant's rows."""
    import torch
    rng = np.random.default_rng(77)
    N, D, k = 6000, 64, 25
    emb = rng.standard_normal((N, D)).astype(np.float32)
    tenants = (np.arange(N) % 3).astype(np.int32)
    temporal = np.where(rng.uniform(size=N) < 0.4, 0.15 * 0.5 ** (rng.uniform(0, 90, N) / 30.0), 0.0)
    post = _sparse_postings(rng, N, 12, 900)
    # the best keyword match of query 0 belongs to ANOTHER tenant: the two normalisations differ
    e2 = eng.__class__(dim=D, device=0)
    try:
        e2.index_load(emb)
        e2.set_tenants(tenants)
        e2.set_temporal(temporal)
        post.load(e2)
        terms_of = [[0, 1], [2, 3, 4], [5], [11, 0, 7]]
        ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
        terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
        q = (emb[[5, 100, 2000, 4001]] + 0.5 * rng.standard_normal((4, D))).astype(np.float32)
        a, b, g = 0.5, 0.4, 0.1
        differs = 0
        for tenant in (1, 2):
            out = e2.hybrid_linear_dev(torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), k, a, b, g,
                                       tenant=tenant)
            torch.cuda.synchronize()
            got = {key: v.cpu().numpy() for key, v in out.items()}
            mine = np.nonzero(tenants == tenant)[0]
            for qi, t in enumerate(terms_of):
                raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, t)
                m = raw[mine].max() if raw[mine].max() > 0 else 1.0
                differs += m != (raw.max() if raw.max() > 0 else 1.0)
                kw = raw[mine] / m
                sem = O.cosine_matrix(q[qi:qi + 1], emb[mine])[0]
                hyb = (a * sem + b * kw) + g * temporal[mine]
                order = O.stable_topk_desc(hyb, k)
                assert got["rows"][qi].tolist() == mine[order].tolist()
                assert got["keyword"][qi].tolist() == kw[order].tolist()
                np.testing.assert_allclose(got["hybrid"][qi], hyb[order], atol=1e-12)
        assert differs > 0                                  # the tenant's max was not the corpus max at least once
    finally:
        e2.close()


def test_fused_second_pass_overflow_after_a_larger_dense_batch(eng):
    """the fused re-emission of rag_hybrid_linear_dev reads the bias row of every one of its 256 query-map
    slots. Slots past the overflowed-query count used to keep whatever an EARLIER search left there - after a 1024-query dense
    search whose overflow list held indices >= 256, that indexed far outside the 256-row bias buffer. Now dead slots map to
    query 0. A 1024-query dense search is forced to overflow everywhere, then a 40-query linear fusion is forced to overflow:
    every query must be recovered by the second pass (no float64 scan) with the oracle's result."""
    import torch
    rng = np.random.default_rng(91)
    N, D, k = 300_000, 64, 100
    emb = rng.standard_normal((N, D)).astype(np.float32)
    post = _sparse_postings(rng, N, 16, 4000)
    temporal = np.where(rng.uniform(size=N) < 0.3, 0.15 * 0.5 ** (rng.uniform(0, 90, N) / 30.0), 0.0)
    e2 = eng.__class__(dim=D, device=0)
    try:
        e2.index_load(emb)
        e2.set_temporal(temporal)
        post.load(e2)
        e2.set_option("stage_growth", 100000)               # one threshold stage over ~298k rows: ~15k keys per query emitted
        qbig = (emb[rng.integers(0, N, 1024)] + 0.5 * rng.standard_normal((1024, D))).astype(np.float32)
        e2.dense_topk(qbig, k)
        st = e2.dense_stats()
        assert st["overflowed"] >= 1024, st                 # the overflow list now holds 256 query indices drawn from 0..1023
        Q = 40
        rows_q = rng.integers(0, N, Q)
        q = (emb[rows_q] + 0.5 * rng.standard_normal((Q, D))).astype(np.float32)
        terms_of = [[int(x) for x in rng.integers(0, 16, int(rng.integers(1, 5)))] for _ in range(Q)]
        ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
        terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
        a, b, g = 0.55, 0.35, 0.10
        out = e2.hybrid_linear_dev(torch.from_numpy(q).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda(), k, a, b, g)
        torch.cuda.synchronize()
        st = e2.dense_stats()
        assert st["overflowed"] >= Q and st["second_pass"] == Q and st["exact_scan"] == 0, st
        got = {key: v.cpu().numpy() for key, v in out.items()}
        for qi in (0, 7, 39):
            raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, terms_of[qi])
            kw = raw / (raw.max() if raw.max() > 0 else 1.0)
            sem = O.cosine_matrix(q[qi:qi + 1], emb)[0]
            hyb = (a * sem + b * kw) + g * temporal
            order = O.stable_topk_desc(hyb, k)
            assert got["rows"][qi].tolist() == order.tolist()
            np.testing.assert_allclose(got["hybrid"][qi], hyb[order], atol=1e-12)
    finally:
        e2.close()
