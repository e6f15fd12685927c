"""Host-side BM25 index builder (optimized-rag_amd/bm25.py::Bm25Postings.from_corpus) against the straightforward
restatement of what rank-bm25 0.2.2 does at the reference's call site (/root/reference/rag/retrieval.py:324-347:
`BM25Okapi([doc.lower().split() for doc in docs])`): per-document frequency dicts, document frequencies in
first-appearance order, postings in document order. The vectorised builder must give the same arrays bit for bit
(term numbering, offsets, docs, tf, document lengths, avgdl, the float64 idf table)."""
import numpy as np
import pytest

from optimized_rag_amd.bm25 import Bm25Postings, tokenize


def _loop_builder(corpus):
    vocab, posting, doc_len = {}, [], []
    for di, text in enumerate(corpus):
        toks = tokenize(text)
        doc_len.append(len(toks))
        freq = {}
        for w in toks:
            freq[w] = freq.get(w, 0) + 1
        for w, f in freq.items():
            if w not in vocab:
                vocab[w] = len(posting)
                posting.append([])
            posting[vocab[w]].append((di, f))
    indptr = np.cumsum([0] + [len(p) for p in posting]).astype(np.int64)
    doc = np.asarray([d for p in posting for d, _ in p], dtype=np.int32)
    tf = np.asarray([f for p in posting for _, f in p], dtype=np.int32)
    return vocab, indptr, doc, tf, np.asarray(doc_len, dtype=np.int32)


def _corpus(rng, n_docs, vocab, max_len):
    words = [f"w{i}" for i in range(vocab)] + ["Mixed", "mixed", "MIXED", "é", "naïve", "x-y", "a.b"]
    return [" ".join(words[int(t)] for t in rng.integers(0, len(words), int(rng.integers(0, max_len + 1)))) for _ in range(n_docs)]


@pytest.mark.parametrize("seed,n_docs,vocab,max_len", [(0, 0, 5, 5), (1, 1, 1, 0), (2, 1, 3, 40), (3, 7, 4, 6), (4, 60, 30, 25),
                                                        (5, 300, 500, 120), (6, 40, 2, 300), (7, 500, 5000, 8)])
def test_from_corpus_equals_the_loop_builder(seed, n_docs, vocab, max_len):
    corpus = _corpus(np.random.default_rng(seed), n_docs, vocab, max_len)
    if n_docs > 2:
        corpus[1] = ""                                            # an empty document in the middle
    p = Bm25Postings.from_corpus(corpus)
    v, indptr, doc, tf, doc_len = _loop_builder(corpus)
    assert list(p.vocab.items()) == list(v.items())               # same numbering, same (first-appearance) order
    for got, exp in ((p.indptr, indptr), (p.doc, doc), (p.tf, tf), (p.doc_len, doc_len)):
        assert got.dtype == exp.dtype
        np.testing.assert_array_equal(got, exp)
    assert p.avgdl == (int(doc_len.sum()) / n_docs if n_docs else 0.0)
    exp_idf = Bm25Postings.idf_table(np.diff(indptr), n_docs) if len(v) else np.zeros(0)
    np.testing.assert_array_equal(p.idf, exp_idf)
    # docs ascending inside every posting list
    for t in range(len(v)):
        seg = p.doc[p.indptr[t]:p.indptr[t + 1]]
        assert (np.diff(seg) > 0).all()


def _idf_loop(df, n_docs, epsilon=0.25):
    import math
    idf = [math.log(n_docs - int(d) + 0.5) - math.log(int(d) + 0.5) for d in df]
    s = 0
    for v in idf:
        s += v
    avg = s / len(idf) if idf else 0.0
    eps = epsilon * avg
    return np.array([eps if v < 0 else v for v in idf], dtype=np.float64)


@pytest.mark.parametrize("seed,n_docs,n_terms", [(0, 1, 1), (1, 3, 50), (2, 100, 3000), (3, 100, 40), (4, 5000, 200), (5, 20000, 30000)])
def test_idf_table_equals_rank_bm25_loop_bit_for_bit(seed, n_docs, n_terms):
    """rank-bm25 0.2.2 `_calc_idf`: math.log per term, `idf_sum += idf` left to right, negatives -> epsilon * average. Both the
    tabulated-logarithm path (small corpus, many terms) and the per-term path; dfs include 1, N and values above N/2 (negative idf)."""
    rng = np.random.default_rng(seed)
    df = rng.integers(1, n_docs + 1, n_terms)
    df[0] = n_docs
    df[-1] = 1
    np.testing.assert_array_equal(Bm25Postings.idf_table(df, n_docs), _idf_loop(df, n_docs))
