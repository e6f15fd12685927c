"""Known-answer vectors for the BM25Okapi restatement (rank-bm25 is absent: parity unpinned upstream).
The expected numbers below are computed by hand from the published formula, independent of the oracle code."""
import math

import numpy as np

from oracle import rag_oracle as O


def test_bm25_hand_computed():
    corpus = ["a b b c", "a d", "b b b b e f", "g"]
    tok = [d.split() for d in corpus]
    bm = O.BM25Okapi(tok)
    N, avgdl = 4, (4 + 2 + 6 + 1) / 4
    assert bm.avgdl == avgdl
    # df: a=2 b=2 c=1 d=1 e=1 f=1 g=1
    idf_a = math.log(N - 2 + 0.5) - math.log(2 + 0.5)          # = 0.0  (not negative -> kept)
    idf_1 = math.log(N - 1 + 0.5) - math.log(1 + 0.5)
    assert bm.idf["a"] == idf_a == 0.0 and bm.idf["c"] == idf_1
    k1, b = 1.5, 0.75

    def term(idf, tf, dl):
        return idf * (tf * (k1 + 1) / (tf + k1 * (1 - b + b * dl / avgdl)))

    s = bm.get_scores(["b", "c", "c", "zzz"])                  # repeated query token counts twice
    exp0 = term(bm.idf["b"], 2, 4) + 2 * term(idf_1, 1, 4)
    exp2 = term(bm.idf["b"], 4, 6)
    np.testing.assert_allclose(s, [exp0, 0.0, exp2, 0.0], rtol=0, atol=1e-15)
    n = O.bm25_scores("b c c zzz", corpus)
    assert max(n) == 1.0 and n[1] == 0.0


def test_bm25_negative_idf_floor():
    # 'x' in 3 of 4 docs -> idf = ln(1.5) - ln(3.5) < 0 -> replaced by epsilon * average_idf
    tok = [["x", "y"], ["x"], ["x", "z"], ["w"]]
    bm = O.BM25Okapi(tok)
    raw = {"x": math.log(1.5) - math.log(3.5), "y": math.log(3.5) - math.log(1.5)}
    raw["z"] = raw["w"] = raw["y"]
    avg = (raw["x"] + raw["y"] + raw["z"] + raw["w"]) / 4       # first-appearance order x,y,z,w
    assert bm.idf["x"] == 0.25 * avg
    assert bm.idf["y"] == raw["y"]


def test_bm25_edge_cases():
    assert O.bm25_scores("q", []) == []
    assert O.bm25_scores("q", ["", "   "]) == [0.0, 0.0]        # retrieval.py:329-331
    assert O.bm25_scores("nothing", ["a b", "c d"]) == [0.0, 0.0]   # max <= 0 -> divide by 1.0


def test_bm25_csr_form_is_bit_identical_to_the_dense_form():
    """oracle.bm25_scores_csr (used by the GPU tests on corpora too large for token dicts) == BM25Okapi.get_scores."""
    import tools_textgen as T
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(5)
    corpus = [T.make_doc(rng, int(rng.integers(1, 5))) for _ in range(300)]
    corpus[10] = ""
    post = Bm25Postings.from_corpus(corpus)
    bm = O.BM25Okapi([O.tokenize(c) for c in corpus])
    for q in ["memory vector index", "paris paris london", "zzz unknown", "kernel"]:
        _, terms = post.encode_queries([q])
        got = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, terms.tolist())
        np.testing.assert_array_equal(got, bm.get_scores(O.tokenize(q)))
