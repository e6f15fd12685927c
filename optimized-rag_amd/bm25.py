"""Host-side BM25 index builder: text -> term-major CSR postings + rank-bm25's float64 idf table.

Tokeniser and statistics follow the reference call site /root/reference/rag/retrieval.py:324-347
(`doc.lower().split()`, BM25Okapi defaults k1=1.5, b=0.75, epsilon=0.25). Scoring runs on the GPU
(csrc/bm25.hip) through rag_bm25_load_host / rag_bm25_topk_host / rag_bm25_scores_host.
"""
import math

import numpy as np

K1, B, EPSILON = 1.5, 0.75, 0.25


def tokenize(text):
    return text.lower().split()


class Bm25Postings:
    """CSR postings (term-major, docs ascending) + idf. Terms are numbered in first-appearance order, which is
    the dict order rank-bm25 sums idf in (the float64 mean depends on that order)."""

    def __init__(self, indptr, doc, tf, doc_len, idf, avgdl, vocab=None, k1=K1, b=B):
        self.indptr, self.doc, self.tf, self.doc_len, self.idf = indptr, doc, tf, doc_len, idf
        self.avgdl, self.vocab, self.k1, self.b = avgdl, vocab, k1, b

    @property
    def n_docs(self):
        return int(self.doc_len.shape[0])

    @staticmethod
    def idf_table(df, n_docs, epsilon=EPSILON):
        """idf = ln(N-df+0.5) - ln(df+0.5); negatives replaced by epsilon * mean(idf) (sequential float64 sum).
        Every logarithm is `math.log` of a half-integer <= N + 0.5 (numpy's log is not guaranteed to round like libm's), so
        for a small corpus with a large vocabulary (the ad-hoc per-call index) the N + 1 possible values are tabulated once
        instead of two calls per term; the mean is the left-to-right running sum (`np.cumsum`), as `sum += idf` is."""
        df = np.asarray(df, dtype=np.int64)
        if df.size == 0:
            return np.zeros(0, dtype=np.float64)
        n_docs = int(n_docs)
        if 0 <= int(df.min()) and int(df.max()) <= n_docs and n_docs + 1 <= 2 * df.size:
            half_log = np.array([math.log(j + 0.5) for j in range(n_docs + 1)], dtype=np.float64)
            idf = half_log[n_docs - df] - half_log[df]
        else:           # large corpus: one pair of logarithms per DISTINCT document frequency (millions of terms share a few thousand)
            udf, inv = np.unique(df, return_inverse=True)
            idf = np.array([math.log(n_docs - int(d) + 0.5) - math.log(int(d) + 0.5) for d in udf], dtype=np.float64)[inv]
        avg = float(np.cumsum(idf)[-1]) / idf.size
        return np.where(idf < 0, epsilon * avg, idf)

    @classmethod
    def from_corpus(cls, corpus, k1=K1, b=B, epsilon=EPSILON):
        """Tokenise and build the CSR. Term numbers are assigned in first-appearance order (dict lookups only, no per-posting Python objects); the
        (term, doc) -> tf counting, the doc-ascending order inside a posting list and the offsets are one sort of the packed
        (term, doc) keys (the ad-hoc `hybrid_search` path builds an index per call: 100 passages of 200 tokens took 27 ms
        with per-posting Python loops, 5 ms this way)."""
        n = len(corpus)
        vocab, ids = {}, []
        doc_len = np.zeros(n, dtype=np.int32)
        for di, text in enumerate(corpus):
            toks = tokenize(text)
            doc_len[di] = len(toks)
            for w in toks:
                if w not in vocab:
                    vocab[w] = len(vocab)
            ids.extend(map(vocab.__getitem__, toks))
        V = len(vocab)
        term_of_tok = np.asarray(ids, dtype=np.int64)
        doc_of_tok = np.repeat(np.arange(n, dtype=np.int64), doc_len)
        key, tf = np.unique(term_of_tok * max(n, 1) + doc_of_tok, return_counts=True)     # sorted: term-major, docs ascending
        indptr = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(np.bincount(key // max(n, 1), minlength=V), out=indptr[1:])
        doc = (key % max(n, 1)).astype(np.int32)
        avgdl = int(doc_len.sum()) / n if n else 0.0
        idf = cls.idf_table(np.diff(indptr), n, epsilon) if V else np.zeros(0)
        return cls(indptr, doc, tf.astype(np.int32), doc_len, idf, avgdl, vocab, k1, b)

    def shard(self, begin, end):
        """Doc-partitioned slice [begin, end) for row-sharded search: postings of the shard's docs
        with LOCAL doc numbers, the GLOBAL idf table / avgdl / vocabulary replicated (BM25 statistics are corpus-wide)."""
        counts = np.diff(self.indptr)
        term_of = np.repeat(np.arange(counts.shape[0], dtype=np.int64), counts)
        keep = (self.doc >= begin) & (self.doc < end)
        indptr = np.zeros(counts.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(term_of[keep], minlength=counts.shape[0]), out=indptr[1:])
        return Bm25Postings(indptr, (self.doc[keep] - begin).astype(np.int32), self.tf[keep].copy(),
                            self.doc_len[begin:end].copy(), self.idf, self.avgdl, self.vocab, self.k1, self.b)

    def encode_queries(self, queries):
        """List[str] -> (term_ptr int32 [Q+1], terms int32) with repeats kept and -1 for unknown tokens."""
        ptr, terms = [0], []
        for q in queries:
            for w in tokenize(q):
                terms.append(self.vocab.get(w, -1))
            ptr.append(len(terms))
        return np.asarray(ptr, dtype=np.int32), np.asarray(terms, dtype=np.int32)

    def load(self, engine):
        engine.bm25_load(self.indptr, self.doc, self.tf, self.doc_len, self.idf, self.avgdl, self.k1, self.b)
        return self
