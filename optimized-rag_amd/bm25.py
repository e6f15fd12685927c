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
        """idf = ln(N-df+0.5) - ln(df+0.5); negatives replaced by epsilon * mean(idf) (sequential float64 sum)."""
        idf = [math.log(n_docs - int(d) + 0.5) - math.log(int(d) + 0.5) for d in df]
        s = 0
        for v in idf:
            s += v
        avg = s / len(idf) if idf else 0.0
        eps = epsilon * avg
        return np.array([eps if v < 0 else v for v in idf], dtype=np.float64)

    @classmethod
    def from_corpus(cls, corpus, k1=K1, b=B, epsilon=EPSILON):
        vocab, posting = {}, []
        doc_len = np.zeros(len(corpus), dtype=np.int32)
        for di, text in enumerate(corpus):
            toks = tokenize(text)
            doc_len[di] = len(toks)
            freq = {}
            for w in toks:
                freq[w] = freq.get(w, 0) + 1
            for w, f in freq.items():
                t = vocab.get(w)
                if t is None:
                    t = vocab[w] = len(posting)
                    posting.append([])
                posting[t].append((di, f))
        V = len(posting)
        indptr = np.zeros(V + 1, dtype=np.int64)
        for t in range(V):
            indptr[t + 1] = indptr[t] + len(posting[t])
        doc = np.empty(int(indptr[-1]), dtype=np.int32)
        tf = np.empty(int(indptr[-1]), dtype=np.int32)
        for t in range(V):
            a = int(indptr[t])
            for j, (d, f) in enumerate(posting[t]):
                doc[a + j] = d
                tf[a + j] = f
        n = len(corpus)
        avgdl = int(doc_len.sum()) / n if n else 0.0
        idf = cls.idf_table(np.diff(indptr), n, epsilon) if V else np.zeros(0)
        return cls(indptr, doc, tf, doc_len, idf, avgdl, vocab, k1, b)

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
