"""Pin the CPU oracle (oracle/rag_oracle.py) against golden vectors produced by running the reference's
own Python (tools/make_golden.py). CPU-only; no GPU, no /root/reference access at run time."""
import json
import os
from datetime import datetime

import numpy as np
import pytest

from oracle import rag_oracle as O

TOL = 1e-12      # float64, different summation order only


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_cosine_all_copies(golden_dir):
    g = np.load(os.path.join(golden_dir, "cosine.npz"))
    a, b = g["a"], g["b"]
    mine = np.array([O.cosine(a[i], b[i]) for i in range(len(a))])
    for k in ("retrieval", "openai", "mmr", "consistency", "compressor", "helpers"):
        np.testing.assert_allclose(mine, g["exp_" + k], rtol=0, atol=TOL, err_msg=k)
    assert mine[5] == 0.0 and mine[6] == 0.0                    # zero-norm -> exactly 0.0
    assert abs(mine[3] - 1.0) < 1e-12 and abs(mine[4] + 1.0) < 1e-12
    M = O.cosine_matrix(a, b)
    np.testing.assert_allclose(np.diag(M), g["exp_retrieval"], atol=TOL)
    assert O.cosine([], [1.0, 2.0], empty_is_zero=True) == g["mmr_empty"][0] == 0.0
    assert O.cosine([1.0], [], empty_is_zero=True) == g["mmr_empty"][1] == 0.0
    assert abs(O.cosine([1.0, 2.0, 3.0], [1.0, 2.0]) - float(g["trunc"])) < TOL


def test_hybrid_search(golden_dir):
    g = load(golden_dir, "hybrid_search.json")
    assert {k: tuple(v[x] for x in ("alpha", "beta", "gamma")) for k, v in g["intent_weights"].items()} == O.INTENT_WEIGHTS
    for c in g["cases"]:
        idx, rows = O.hybrid_search(
            c["query"], c["corpus"], np.array(c["embeddings"], dtype=np.float32),
            np.array(c["query_embedding"], dtype=np.float32), top_k=c["top_k"], metadata=c["metadata"],
            intent=c["intent"], default_weights=tuple(c["alpha_beta_gamma_default"]),
            use_adaptive_weights=c["use_adaptive_weights"], bm25_available=False,
            now=datetime.fromisoformat(c["now"]))
        assert idx == c["expected_idx"]
        for r, e in zip(rows, c["expected"]):
            for k in e:
                assert abs(r[k] - e[k]) < TOL, k
    for kcase in g["keyword"]:
        assert O.simple_keyword_scores(kcase["query"], kcase["corpus"]) == kcase["expected"]


def test_rrf(golden_dir):
    g = load(golden_dir, "rrf.json")
    for c in g["cases"]:
        keys, scores, ranks = O.rrf_fuse(c["lists"], k=c["k"], top_k=c["top_k"])
        assert keys == c["expected_ids"]
        assert scores == c["expected_scores"]                   # same fp64 ops in the same order: bit-exact
        for key, rk in zip(keys, ranks):
            for li, lst in enumerate(c["lists"]):
                assert rk[li] == (lst.index(key) + 1 if key in lst else 0)
    d = g["dup"]
    lists = [[x if x is not None else "" for x in l] for l in d["lists"]]
    keys, scores, _ = O.rrf_fuse(lists, k=60, top_k=10)
    assert keys == d["expected_contents"] and scores == d["expected_scores"]


def test_mmr(golden_dir):
    g = load(golden_dir, "mmr.json")
    for c in g["class"]:
        pos, sc = O.mmr_class(c["q"], c["emb"], c["top_k"], c["lambda"])
        assert pos == c["expected_pos"]
        np.testing.assert_allclose(sc, c["expected_mmr"], atol=TOL)
    for c in g["helper"]:
        assert O.mmr_helper(c["q"], c["emb"], c["k"], c["lambda"]) == c["expected_pos"]


def test_consistency(golden_dir):
    g = load(golden_dir, "consistency.json")
    for c in g["cases"]:
        table = c["embeddings"]
        for d, exp in zip(c["docs"], c["expected_claims"]):
            assert O.extract_claims(d["content"]) == exp
        out = O.check_consistency(c["docs"], lambda texts: [table[t] for t in texts], threshold=c["threshold"])
        exp = c["expected"]
        assert out["consistent"] == exp["consistent"]
        assert out["contradiction_count"] == exp["contradiction_count"]
        assert out["total_claims"] == exp["total_claims"]
        assert abs(out["confidence"] - exp["confidence"]) < TOL
        assert out["warning"] == exp["warning"]
        assert out["contradictions"] == exp["contradictions"]
    e = g["edge"]
    assert O.check_consistency([{"content": "x"}], None) == e["one_doc"]
    assert O.check_consistency([{"content": "Tiny."}, {"content": "Also tiny."}], None) == e["few_claims"]

    def boom(t):
        raise RuntimeError("down")

    out = O.check_consistency(e["embed_fail_docs"], boom)
    assert out == e["embed_fail"]
    for p in e["is_contradiction"]:
        assert O.is_contradiction(p["a"], p["b"]) == p["expected"]


def test_compressor_pieces(golden_dir):
    g = load(golden_dir, "compressor.json")
    sh = g["score_hybrid"]
    table = sh["embeddings"]
    mine = O.score_sentences_hybrid(sh["query"], sh["sentences"], table[sh["query"]], [table[s] for s in sh["sentences"]])
    np.testing.assert_allclose(mine, sh["expected"], atol=TOL)
    for c in g["lexical"]:
        assert abs(O.score_sentence_lexical(c["q"], c["s"]) - c["expected"]) < TOL
    for c in g["split"]:
        assert O.split_sentences(c["text"]) == c["expected"]


def test_reranker_postprocessing(golden_dir):
    g = load(golden_dir, "rerankers.json")
    oai = g["openai"]
    emb = np.array(oai["emb"], dtype=np.float32)
    origs = [(r.get("similarity", 0) or r.get("score", 0)) for r in oai["results"]]
    sc = O.openai_rerank_scores(emb[0], emb[1:], origs)
    order = O.stable_topk_desc(sc, oai["top_k"])
    assert [int(i) for i in order] == oai["expected_pos"]
    np.testing.assert_allclose([sc[i] for i in order], oai["expected_rerank"], atol=TOL)
    cr = g["cross"]
    sig = [O.sigmoid(float(np.float32(x))) for x in cr["logits"][:-1]]      # last one (-30) is still finite
    sig.append(O.sigmoid(float(np.float32(cr["logits"][-1]))))
    order = O.stable_topk_desc(sig, cr["top_k"])
    assert [int(i) for i in order] == [e["pos"] for e in cr["expected"]]
    for i, e in zip(order, cr["expected"]):
        assert abs(sig[i] - e["cross_encoder_score"]) < 1e-15
