#!/usr/bin/env python3
"""Generate golden input/output vectors for the hot path by RUNNING the reference's own
pure-Python code in this container.

Only data (inputs + expected outputs) is written to tests/golden/. No reference source is
copied; the reference modules are loaded from /root/reference by file path under private
names, with namespace stubs for the heavy packages they would otherwise pull in.

Run once, here (the GPU box has no /root/reference):   python tools/make_golden.py
"""
import datetime as _real_datetime
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FROZEN_NOW = _real_datetime.datetime(2026, 3, 27, 12, 0, 0)   # naive, like datetime.now()


# --------------------------------------------------------------------------------------
# reference loading (recipe: SURVEY Appendix D)
# --------------------------------------------------------------------------------------
def _load_by_path(private_name, relpath):
    spec = importlib.util.spec_from_file_location(private_name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _install_stubs():
    cfg = types.ModuleType("config")
    cfg.ENABLE_TEMPORAL_BOOST = True          # config.py:38,131
    cfg.RECENCY_WEIGHT = 0.15                 # config.py:39,132
    cfg.RECENCY_HALF_LIFE_DAYS = 30           # config.py:40,133
    cfg.COMPRESSION_MIN_THRESHOLD = 0.005     # config.py:215
    cfg.COMPRESSION_INTENT_THRESHOLDS = {     # config.py:216-221
        "QUESTION_ANSWERING": 0.25, "SEARCH": 0.2, "CONVERSATIONAL": 0.15, "MULTI_HOP_REASONING": 0.3}
    cfg.MIN_QUALITY_SCORE = 0.5               # config.py:189
    cfg.MIN_AVG_RELEVANCE_SCORE = 0.35        # config.py:192
    sys.modules["config"] = cfg

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("rag", os.path.join(REF, "rag"))
    ns("rag.models", os.path.join(REF, "rag", "models"))
    ns("rag.nodes", os.path.join(REF, "rag", "nodes"))
    ns("memory", os.path.join(REF, "memory"))
    ns("prompts", os.path.join(REF, "prompts"))
    emb = types.ModuleType("memory.embeddings")

    class EmbeddingService:                   # placeholder type for annotations only
        pass

    emb.EmbeddingService = EmbeddingService
    sys.modules["memory.embeddings"] = emb
    ld = types.ModuleType("langdetect")
    ld.detect = lambda s: "en"
    lde = types.ModuleType("langdetect.lang_detect_exception")

    class LangDetectException(Exception):
        pass

    lde.LangDetectException = LangDetectException
    sys.modules["langdetect"] = ld
    sys.modules["langdetect.lang_detect_exception"] = lde


class _FrozenClock:
    """Context manager: `from datetime import datetime` inside hybrid_search sees a frozen now()."""

    def __enter__(self):
        real = _real_datetime

        class FrozenDT(real.datetime):
            @classmethod
            def now(cls, tz=None):
                return FROZEN_NOW

        fake = types.ModuleType("datetime")
        fake.datetime = FrozenDT
        fake.timedelta = real.timedelta
        self._saved = sys.modules["datetime"]
        sys.modules["datetime"] = fake
        return self

    def __exit__(self, *a):
        sys.modules["datetime"] = self._saved


# --------------------------------------------------------------------------------------
# synthetic data
# --------------------------------------------------------------------------------------
WORDS = ("system memory vector index query document retrieval ranking fusion agent graph node "
         "embedding cosine score keyword search context token model latency cache batch shard "
         "kernel bandwidth matrix tile stream buffer policy storage engine network protocol "
         "database table column record update delete insert commit branch merge release "
         "Paris London Berlin Madrid Rome Lisbon Vienna Prague Dublin Oslo").split()


def make_sentence(rng, n_lo=6, n_hi=14, end="."):
    n = int(rng.integers(n_lo, n_hi))
    w = [WORDS[int(i)] for i in rng.integers(0, len(WORDS), n)]
    w[0] = w[0].capitalize()
    return " ".join(w) + end


def make_doc(rng, n_sent):
    return " ".join(make_sentence(rng) for _ in range(n_sent))


class FakeEmbeddingService:
    """Seeded text->vector map; records every text it embedded so fixtures can carry the table."""

    def __init__(self, dim, seed):
        self.dim, self.seed, self.table = dim, seed, {}

    def _vec(self, text):
        if text not in self.table:
            h = abs(hash_str(text)) % (2**31)
            rng = np.random.default_rng([self.seed, h])
            v = rng.standard_normal(self.dim)
            self.table[text] = [float(x) for x in v.astype(np.float32)]
        return list(self.table[text])

    def generate_embedding(self, text):
        return self._vec(text)

    def generate_embeddings_batch(self, texts):
        return [self._vec(t) for t in texts]


def hash_str(s):
    h = 1469598103934665603
    for ch in s.encode("utf-8"):
        h = ((h ^ ch) * 1099511628211) % (1 << 64)
    return h


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def tolist64(a):
    """float32 array -> python floats (the reference computes in float64 from these)."""
    return [[float(v) for v in row] for row in np.asarray(a)]


def dump_json(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


# --------------------------------------------------------------------------------------
def gen_cosine(retrieval, reranker, consistency, compressor, helpers):
    rng = np.random.default_rng(101)
    D = 1536
    a = rng.standard_normal((24, D)).astype(np.float32)
    b = rng.standard_normal((24, D)).astype(np.float32)
    b[3] = a[3]                # identical -> 1.0
    b[4] = -a[4]               # opposite  -> -1.0
    a[5] = 0.0                 # zero norm -> 0.0   (retrieval.py:368-369)
    b[6] = 0.0
    b[7] = a[7] * 3.5          # scale invariance
    b[8] = a[8] + 1e-3 * rng.standard_normal(D).astype(np.float32)   # near-duplicate
    hr = retrieval.HybridRetriever.__new__(retrieval.HybridRetriever)
    oai = reranker.OpenAIReranker.__new__(reranker.OpenAIReranker)
    mmr = reranker.MMRDiversifier(0.7)
    cc = consistency.ConsistencyChecker.__new__(consistency.ConsistencyChecker)
    cp = compressor.ContextCompressor.__new__(compressor.ContextCompressor)
    exp = {k: [] for k in ("retrieval", "openai", "mmr", "consistency", "compressor", "helpers")}
    for i in range(len(a)):
        va, vb = [float(x) for x in a[i]], [float(x) for x in b[i]]
        exp["retrieval"].append(hr._cosine_similarity(va, vb))
        exp["openai"].append(oai._cosine_similarity(va, vb))
        exp["mmr"].append(mmr._cosine_similarity(va, vb))
        exp["consistency"].append(cc._cosine_similarity(va, vb))
        exp["compressor"].append(cp._cosine_similarity(va, vb))
        exp["helpers"].append(helpers.cosine_similarity(va, vb))
    # the MMR copy returns 0.0 on empty vectors (reranker.py:199-200)
    mmr_empty = [mmr._cosine_similarity([], [1.0, 2.0]), mmr._cosine_similarity([1.0], [])]
    # zip() truncation on dimension mismatch (Appendix B.1)
    trunc = hr._cosine_similarity([1.0, 2.0, 3.0], [1.0, 2.0])
    np.savez_compressed(os.path.join(OUT, "cosine.npz"), a=a, b=b,
                        mmr_empty=np.array(mmr_empty), trunc=np.array(trunc),
                        **{"exp_" + k: np.array(v, dtype=np.float64) for k, v in exp.items()})
    print("wrote cosine.npz")


def gen_hybrid_search(retrieval):
    rng = np.random.default_rng(202)
    cases = []
    hr = retrieval.HybridRetriever(memory_manager=None, document_store=None, agent_id="agent-x")
    assert hr.bm25_available is False, "rank_bm25 is not installed here: keyword fallback path is what runs"
    D = 96
    for ci, (n, intent, with_meta, top_k, adaptive) in enumerate([
            (12, "search", False, 5, True),
            (40, "question_answering", True, 10, True),
            (40, None, True, 7, True),
            (25, "Multi Hop Reasoning", False, 25, True),
            (25, "unknown_intent", True, 30, True),
            (18, "summarization", True, 6, False),
            (1, "search", False, 3, True),
    ]):
        corpus = [make_doc(rng, int(rng.integers(1, 4))) for _ in range(n)]
        if n > 5:
            corpus[3] = corpus[2]                      # duplicate doc -> exact tie, stable order
        emb = unit_rows(rng, n, D)
        if n > 5:
            emb[3] = emb[2]
        q_words = [WORDS[int(i)] for i in rng.integers(0, len(WORDS), 5)]
        query = " ".join(q_words)
        qe = (emb[int(rng.integers(0, n))] + 0.5 * rng.standard_normal(D)).astype(np.float32)
        meta = None
        if with_meta:
            meta = []
            for i in range(n):
                days = float(rng.uniform(0, 200))
                ts = (FROZEN_NOW - _real_datetime.timedelta(days=days)).isoformat()
                r = i % 5
                if r == 0:
                    meta.append({"created_at": ts})
                elif r == 1:
                    meta.append({"uploaded_at": ts})
                elif r == 2:
                    meta.append({})                       # no timestamp -> 0.0
                elif r == 3:
                    meta.append({"created_at": "not-a-date"})   # ValueError -> 0.0
                else:
                    meta.append({"created_at": ts, "uploaded_at": "ignored"})
        hr.use_adaptive_weights = adaptive
        with _FrozenClock():
            out = hr.hybrid_search(query=query, corpus=corpus, embeddings=tolist64(emb),
                                   query_embedding=[float(x) for x in qe], top_k=top_k,
                                   documents_metadata=meta, query_intent=intent)
        # identify each returned doc by its position in the input corpus (first unused match)
        used, idx = set(), []
        for r in out:
            for i in range(n):
                if i not in used and corpus[i] == r["content"] and r["embedding"] == tolist64(emb[i:i + 1])[0]:
                    # duplicates: stable sort keeps input order
                    used.add(i)
                    idx.append(i)
                    break
        assert len(idx) == len(out)
        cases.append({
            "query": query, "corpus": corpus, "embeddings": emb.tolist(), "query_embedding": qe.tolist(),
            "top_k": top_k, "metadata": meta, "intent": intent, "use_adaptive_weights": adaptive,
            "now": FROZEN_NOW.isoformat(),
            "alpha_beta_gamma_default": [hr.alpha, hr.beta, hr.gamma],
            "expected_idx": idx,
            "expected": [{k: r[k] for k in ("hybrid_score", "semantic_score", "keyword_score", "temporal_score")}
                         for r in out],
            "expected_has_metadata": ["metadata" in r for r in out],
        })
    # simple keyword scores, standalone (retrieval.py:349-360), incl. empty query
    kw = []
    for q in ["memory vector index", "", "Memory  MEMORY memory", "zzz"]:
        corp = ["memory vector", "Index of the vector memory index", "", "nothing here"]
        kw.append({"query": q, "corpus": corp, "expected": hr._simple_keyword_scores(q, corp)})
    dump_json("hybrid_search.json", {"cases": cases, "keyword": kw,
                                     "intent_weights": retrieval.HybridRetriever.INTENT_WEIGHTS})


def gen_rrf(reranker):
    rng = np.random.default_rng(303)
    cases = []
    for (L, lens, k, top_k) in [(2, (5, 5), 60, 10), (2, (100, 100), 60, 20), (3, (30, 50, 10), 60, 15),
                                (2, (8, 0), 60, 10), (1, (12,), 10, 5), (2, (100, 100), 1, 200)]:
        pool = [f"doc-{i}" for i in range(150)]
        lists = []
        for ln in lens:
            ids = rng.permutation(len(pool))[:ln]
            lists.append([{"content": pool[int(i)], "id": int(i)} for i in ids])
        fused = reranker.ReciprocalRankFusion(k=k).fuse([[dict(d) for d in l] for l in lists], top_k=top_k)
        cases.append({"k": k, "top_k": top_k, "lists": [[d["id"] for d in l] for l in lists],
                      "expected_ids": [d["id"] for d in fused],
                      "expected_scores": [d["rrf_score"] for d in fused]})
    # duplicate content inside one list and docs without content (Appendix B.6)
    l1 = [{"content": "x"}, {"content": "y"}, {"content": "x"}, {}]
    l2 = [{"content": "y"}, {}, {"content": "z"}]
    fused = reranker.ReciprocalRankFusion(k=60).fuse([l1, l2], top_k=10)
    dup = {"lists": [[d.get("content") for d in l1], [d.get("content") for d in l2]],
           "expected_contents": [d.get("content", "") for d in fused],
           "expected_scores": [d["rrf_score"] for d in fused]}
    dump_json("rrf.json", {"cases": cases, "dup": dup})


def gen_mmr(reranker, helpers):
    rng = np.random.default_rng(404)
    D = 128
    cls_cases, helper_cases = [], []
    for (n, top_k, lam) in [(12, 5, 0.7), (30, 8, 0.5), (6, 10, 0.7), (20, 5, 1.0), (20, 5, 0.0)]:
        base = unit_rows(rng, 4, D)
        emb = np.stack([base[int(rng.integers(0, 4))] + 0.4 * rng.standard_normal(D) for _ in range(n)]).astype(np.float32)
        if n >= 12:
            emb[5] = emb[1]                      # exact duplicate embedding -> tie, first max wins
        qe = (base[0] + 0.3 * rng.standard_normal(D)).astype(np.float32)
        docs = [{"content": f"d{i}", "embedding": [float(x) for x in emb[i]], "pos": i} for i in range(n)]
        out = reranker.MMRDiversifier(lam).diversify([float(x) for x in qe], [dict(d) for d in docs], top_k=top_k)
        cls_cases.append({"emb": emb.tolist(), "q": qe.tolist(), "top_k": top_k, "lambda": lam,
                          "expected_pos": [d["pos"] for d in out],
                          "expected_mmr": [d["mmr_score"] for d in out]})
    # invalid-embedding filter (reranker.py:138-151)
    docs = [{"content": "a", "embedding": [1.0, 0.0]}, {"content": "b", "embedding": []},
            {"content": "c", "embedding": [float("nan"), 1.0]}, {"content": "d"},
            {"content": "e", "embedding": [0.6, 0.8]}, {"content": "f", "embedding": [float("inf"), 1.0]},
            {"content": "g", "embedding": (1.0, 0.0)}]
    out = reranker.MMRDiversifier(0.7).diversify([1.0, 0.2], [dict(d) for d in docs], top_k=5)
    invalid = {"expected_contents": [d["content"] for d in out], "expected_mmr": [d["mmr_score"] for d in out]}
    out2 = reranker.MMRDiversifier(0.7).diversify([1.0, 0.2], [{"content": "b", "embedding": []}, {"content": "d"}], top_k=1)
    invalid["none_valid_contents"] = [d["content"] for d in out2]

    for (n, k, lam) in [(12, 5, 0.7), (30, 8, 0.5), (5, 5, 0.7), (4, 9, 0.7), (20, 5, 0.3)]:
        base = unit_rows(rng, 3, D)
        emb = np.stack([base[int(rng.integers(0, 3))] + 0.4 * rng.standard_normal(D) for _ in range(n)]).astype(np.float32)
        qe = (base[1] + 0.3 * rng.standard_normal(D)).astype(np.float32)

        class Svc:
            def generate_embedding(self, text):
                assert text == "the query"
                return [float(x) for x in qe]

        docs = [{"content": f"d{i}", "embedding": [float(x) for x in emb[i]], "pos": i} for i in range(n)]
        out = helpers.apply_mmr("the query", docs, lam, k, Svc())
        helper_cases.append({"emb": emb.tolist(), "q": qe.tolist(), "k": k, "lambda": lam,
                             "expected_pos": [d["pos"] for d in out]})
    dump_json("mmr.json", {"class": cls_cases, "invalid": invalid, "helper": helper_cases})


def gen_consistency(consistency):
    rng = np.random.default_rng(505)
    D = 64
    cases = []
    for ci in range(4):
        svc = FakeEmbeddingService(D, seed=900 + ci)
        docs = []
        for di in range(int(rng.integers(2, 6))):
            sents = [make_sentence(rng) for _ in range(int(rng.integers(2, 6)))]
            sents.append("It is short.")                          # < 20 chars -> dropped
            sents.append("This is a meta statement about nothing at all.")   # meta pattern -> dropped
            docs.append({"content": " ".join(sents), "source": f"src{di}"} if di % 2 == 0 else {"content": " ".join(sents)})
        # plant contradicting near-duplicate claims across docs with (a) negation (b) numbers
        c1 = "The storage engine is designed for low latency reads of 15 records"
        c2 = "The storage engine is not designed for low latency reads of 15 records"
        c3 = "The storage engine was built in 2019 with 40 shards in total"
        c4 = "The storage engine was built in 2021 with 40 shards in total"
        c5 = "The retrieval graph keeps every node inside the same memory shard"
        docs[0]["content"] += " " + c1 + ". " + c3 + ". " + c5 + "."
        docs[1]["content"] += " " + c2 + ". " + c4 + ". " + c5 + "."
        base = svc._vec(c1)
        # make planted pairs highly similar; similarity of (c5,c5) is exactly 1.0
        svc.table[c2] = [float(np.float32(x + 0.02 * rng.standard_normal())) for x in base]
        b3 = svc._vec(c3)
        svc.table[c4] = [float(np.float32(x + 0.6 * rng.standard_normal())) for x in b3] if ci % 2 else \
            [float(np.float32(x + 0.05 * rng.standard_normal())) for x in b3]
        thr = [0.85, 0.85, 0.5, 0.99][ci]
        chk = consistency.ConsistencyChecker(svc, similarity_threshold=thr)
        claims = [chk._extract_claims(d["content"]) for d in docs]
        out = chk.check_consistency([dict(d) for d in docs], "some query")
        cases.append({"docs": docs, "threshold": thr, "embeddings": svc.table, "expected_claims": claims,
                      "expected": out})
    chk = consistency.ConsistencyChecker(FakeEmbeddingService(D, 1), 0.85)
    edge = {"one_doc": chk.check_consistency([{"content": make_doc(rng, 3)}], "q"),
            "few_claims": chk.check_consistency([{"content": "Tiny."}, {"content": "Also tiny."}], "q")}

    class Boom:
        def generate_embeddings_batch(self, t):
            raise RuntimeError("embedding backend down")

    edge["embed_fail"] = consistency.ConsistencyChecker(Boom(), 0.85).check_consistency(
        [{"content": make_doc(np.random.default_rng(1), 3)}, {"content": make_doc(np.random.default_rng(2), 3)}], "q")
    edge["embed_fail_docs"] = [{"content": make_doc(np.random.default_rng(1), 3)}, {"content": make_doc(np.random.default_rng(2), 3)}]
    pairs = [("The cache is not warm", "The cache is warm"), ("There are 12 shards", "There are 14 shards"),
             ("Alpha beta gamma", "Alpha beta gamma delta"), ("It will always work", "It will never work"),
             ("There are 12 shards", "There are 12 shards")]
    edge["is_contradiction"] = [{"a": a, "b": b, "expected": chk._is_contradiction(a, b)} for a, b in pairs]
    dump_json("consistency.json", {"cases": cases, "edge": edge})


def gen_compressor(compressor):
    from rag.models.intent_analysis import QueryIntent
    rng = np.random.default_rng(606)
    D = 64
    cases = []
    for ci, (ndocs, conf, spd, max_tokens, conservative) in enumerate([
            (9, 1.0, 3, 4000, False), (10, 0.7, 2, 4000, False), (8, 0.5, 3, 900, False),
            (9, 0.9, 3, 50, True), (9, 0.9, 3, 100000, True), (5, 1.0, 3, 4000, False), (12, 0.95, 4, 4000, False)]):
        svc = FakeEmbeddingService(D, seed=700 + ci)
        query = " ".join(WORDS[int(i)] for i in rng.integers(0, len(WORDS), 4)) + " the of"
        docs = []
        for di in range(ndocs):
            content = make_doc(rng, int(rng.integers(4, 9)))
            if di == 1:
                content += " " + query.capitalize() + " appears verbatim in this sentence right here."
            d = {"content": content, "filename": f"f{di}.txt"}
            if di % 3 != 2:
                d["score"] = float(np.round(rng.uniform(0.0, 1.0), 3))
            docs.append(d)
        comp = compressor.ContextCompressor(max_tokens=max_tokens, sentences_per_doc=spd,
                                            embedding_service=svc, conservative_mode=conservative)
        out = comp.compress(query, [dict(d) for d in docs], query_intent=QueryIntent.QUESTION_ANSWERING, confidence=conf)
        stats = comp.get_compression_stats(out)
        cases.append({"query": query, "docs": docs, "confidence": conf, "sentences_per_doc": spd,
                      "max_tokens": max_tokens, "conservative": conservative, "embeddings": svc.table,
                      "expected": out, "expected_stats": stats})
    # sentence scoring alone (context_compressor.py:217-241), and lexical scoring (:265-286)
    svc = FakeEmbeddingService(D, seed=777)
    comp = compressor.ContextCompressor(embedding_service=svc, conservative_mode=False)
    sents = [make_sentence(rng, end="") for _ in range(20)]
    q = "vector index query of the system"
    scored = comp._score_sentences_hybrid(q, sents)
    lex = [{"q": qq, "s": ss, "expected": comp._score_sentence_lexical(qq, ss)} for qq, ss in [
        (q, sents[0]), ("the of and", sents[1]), ("index", "The INDEX, index; re-index!"),
        ("vector index", "a vector index is here and vector index again"), ("", "anything")]]
    split = [{"text": t, "expected": comp._split_sentences(t)} for t in [
        docs[0]["content"], "", "Short. Tiny! This one is long enough to be kept? Yes it is long enough as well.  x"]]
    nosvc = compressor.ContextCompressor(sentences_per_doc=2, embedding_service=None, conservative_mode=False)
    docs9 = [{"content": make_doc(rng, 5), "score": 0.9} for _ in range(9)]
    out_lex = nosvc.compress("memory vector index", [dict(d) for d in docs9])
    dump_json("compressor.json", {"cases": cases, "score_hybrid": {"query": q, "sentences": sents,
              "embeddings": svc.table, "expected": [s for _, s in scored]}, "lexical": lex, "split": split,
              "lexical_only": {"docs": docs9, "query": "memory vector index", "expected": out_lex}})


def gen_rerankers(reranker):
    rng = np.random.default_rng(808)
    D = 48
    # OpenAIReranker with a fake client (reranker.py:28-90)
    n = 9
    emb = unit_rows(rng, n + 1, D)

    class Item:
        def __init__(self, e):
            self.embedding = e

    class Resp:
        def __init__(self, data):
            self.data = data

    class Embeddings:
        def create(self, input, model):
            assert len(input) == n + 1
            return Resp([Item([float(x) for x in emb[i]]) for i in range(n + 1)])

    class Client:
        embeddings = Embeddings()

    results = []
    for i in range(n):
        d = {"content": f"doc {i} " + "x" * (9000 if i == 2 else 10), "pos": i}
        if i % 3 == 0:
            d["similarity"] = float(np.round(rng.uniform(0, 1), 3))
        if i % 3 == 1:
            d["score"] = float(np.round(rng.uniform(0, 1), 3))
        if i == 6:
            d["similarity"] = 0           # falls through to score (Appendix B.9)
            d["score"] = 0.77
        results.append(d)
    rr = reranker.OpenAIReranker(Client(), "text-embedding-3-large")
    out = rr.rerank("q", [dict(d) for d in results], top_k=5)
    oai = {"emb": emb.tolist(), "results": results, "top_k": 5,
           "expected_pos": [d["pos"] for d in out], "expected_rerank": [d["rerank_score"] for d in out]}

    class BadClient:
        class embeddings:
            @staticmethod
            def create(input, model):
                raise RuntimeError("api down")

    out_fail = reranker.OpenAIReranker(BadClient(), "m").rerank("q", [dict(d) for d in results], top_k=4)
    oai["fail_pos"] = [d["pos"] for d in out_fail]

    # CrossEncoderReranker post-processing with an injected fake model (reranker.py:320-384).
    ce = reranker.CrossEncoderReranker(model_name="cross-encoder/ms-marco-MiniLM-L-6-v2")
    assert ce.model is None and ce.is_available() is False     # sentence-transformers absent here
    docs = [{"content": "c" * (2500 if i == 1 else 30 + i), "pos": i} for i in range(10)]
    for i in (0, 3, 4):
        docs[i]["score"] = 0.1 * i + 0.05
    docs[4]["embedding_score"] = 0.999                        # already saved -> not overwritten
    fallback = ce.rerank("q", [dict(d) for d in docs], top_k=3)
    logits = np.array([1.5, -2.25, 0.0, 7.75, -9.5, 1.5, 3.125, -0.5, 12.0, -30.0], dtype=np.float32)
    seen = {}

    class FakeModel:
        def predict(self, pairs):
            seen["pairs_len"] = [len(p[1]) for p in pairs]
            seen["q"] = [p[0] for p in pairs]
            return logits

    ce.model = FakeModel()
    out = ce.rerank("the query", [dict(d) for d in docs], top_k=6)
    cross = {"docs": docs, "logits": logits.tolist(), "top_k": 6,
             "fallback_pos": [d["pos"] for d in fallback],
             "pairs_len": seen["pairs_len"],
             "expected": [{k: d.get(k) for k in ("pos", "score", "cross_encoder_score", "cross_encoder_raw_score",
                                                 "embedding_score")} for d in out]}

    class Overflow:
        def predict(self, pairs):
            return np.array([-800.0] * len(pairs))            # math.exp overflow -> except -> results[:top_k]

    ce.model = Overflow()
    cross["overflow_pos"] = [d["pos"] for d in ce.rerank("q", [dict(d) for d in docs], top_k=2)]
    dump_json("rerankers.json", {"openai": oai, "cross": cross})


def gen_bert():
    """Cross-encoder forward oracle: transformers (container code, NOT reference code) BertForSequenceClassification
    at the ms-marco-MiniLM-L-6-v2 shape with seeded-numpy weights. Weights are regenerated
    from the seed by tests; only ids and logits are stored."""
    import torch
    from transformers import BertConfig, BertForSequenceClassification
    sys.path.insert(0, os.path.dirname(OUT.rstrip("/")).rsplit("/tests", 1)[0])
    from oracle.bert_oracle import minilm_config, seeded_weights
    cfg = minilm_config()
    w = seeded_weights(cfg, seed=2024)
    hf = BertForSequenceClassification(BertConfig(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
        num_attention_heads=cfg["heads"], intermediate_size=cfg["ffn"], max_position_embeddings=cfg["max_pos"],
        type_vocab_size=2, hidden_act="gelu", layer_norm_eps=1e-12, num_labels=1)).eval()
    sd = hf.state_dict()
    for k, v in w.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))
    rng = np.random.default_rng(11)
    P, L = 12, 96
    lens = rng.integers(20, L + 1, P)
    lens[0] = L
    lens[1] = 3
    ids = np.zeros((P, L), dtype=np.int64)
    tt = np.zeros((P, L), dtype=np.int64)
    for p in range(P):
        n = int(lens[p])
        ids[p, :n] = rng.integers(1000, cfg["vocab_size"], n)
        ids[p, 0] = 101
        qlen = min(10, n - 1)
        ids[p, qlen] = 102
        ids[p, n - 1] = 102
        tt[p, qlen + 1:n] = 1
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        logits = hf(input_ids=torch.from_numpy(ids), token_type_ids=torch.from_numpy(tt),
                    attention_mask=torch.from_numpy(mask)).logits[:, 0].double().numpy()
    np.savez_compressed(os.path.join(OUT, "bert_minilm.npz"), seed=2024, input_ids=ids.astype(np.int32),
                        token_type_ids=tt.astype(np.int32), lens=lens.astype(np.int32), logits=logits)
    print("wrote bert_minilm.npz", logits[:4])


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    retrieval = _load_by_path("_ref_retrieval", "rag/retrieval.py")
    reranker = _load_by_path("_ref_reranker", "rag/reranker.py")
    consistency = importlib.import_module("rag.consistency_checker")
    compressor = importlib.import_module("rag.context_compressor")
    helpers = importlib.import_module("rag.nodes.helpers")
    gen_cosine(retrieval, reranker, consistency, compressor, helpers)
    gen_hybrid_search(retrieval)
    gen_rrf(reranker)
    gen_mmr(reranker, helpers)
    gen_consistency(consistency)
    gen_compressor(compressor)
    gen_rerankers(reranker)
    if "--no-bert" not in sys.argv:
        gen_bert()


if __name__ == "__main__":
    main()
