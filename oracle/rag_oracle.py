"""TEST INFRASTRUCTURE — CPU oracle (numpy float64 restatement) of the reference's hybrid-retrieval +
rerank hot path. Not product code: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module; the product package never does.

Every function cites the reference file:line (under /root/reference) whose behaviour it restates.

Pinning status
  * cosine, hybrid_search (linear fusion, keyword-overlap path, temporal decay), RRF, both MMR variants,
    consistency checker, context compressor, OpenAI/cross-encoder re-ranker post-processing:
    PINNED by tests/golden/*.json|npz, produced by tools/make_golden.py running the reference's own
    Python in the build container (tests/test_oracle_golden.py).
  * BM25Okapi: third-party `rank-bm25` (requirements.txt:22, pinned only as >=0.2.2) is absent from the
    reference tree and from this image -> restated from the published algorithm (0.2.2); PARITY UNPINNED
    upstream, anchored by hand-computed known-answer vectors (tests/test_oracle_bm25.py).
  * pgvector `<=>` exact scan: extension absent -> restated as exact float64 cosine scan; pinned only
    through the cosine fixtures (same formula). PARITY UNPINNED against a live Postgres.
"""
import math
import re
from datetime import datetime

import numpy as np

# ---------------------------------------------------------------------------------------------
# a1  cosine   (rag/retrieval.py:362-371 and its copies reranker.py:92-101,197-209,
#               consistency_checker.py:241-261, context_compressor.py:243-263, nodes/helpers.py:263-290)
# ---------------------------------------------------------------------------------------------


def cosine(v1, v2, empty_is_zero=False):
    """dot/(|a||b|) in float64; 0.0 when either norm is 0; zip() truncation to the shorter vector."""
    if empty_is_zero and (len(v1) == 0 or len(v2) == 0):        # MMR copy, reranker.py:199-200
        return 0.0
    a = np.asarray(v1, dtype=np.float64)
    b = np.asarray(v2, dtype=np.float64)
    n = min(len(a), len(b))
    dot = float(np.sum(a[:n] * b[:n]))                          # zip truncates the dot product only
    m1 = math.sqrt(float(np.sum(a * a)))
    m2 = math.sqrt(float(np.sum(b * b)))
    if m1 == 0 or m2 == 0:
        return 0.0
    return dot / (m1 * m2)


def cosine_matrix(A, B):
    """[m,D] x [n,D] -> [m,n] float64 cosines, 0.0 where a norm is 0. Row-position independent."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    na = np.sqrt(np.sum(A * A, axis=1))
    nb = np.sqrt(np.sum(B * B, axis=1))
    out = np.empty((A.shape[0], B.shape[0]), dtype=np.float64)
    for i in range(A.shape[0]):
        out[i] = np.sum(B * A[i][None, :], axis=1)
    den = na[:, None] * nb[None, :]
    return np.where(den == 0, 0.0, out / np.where(den == 0, 1.0, den))


def stable_topk_desc(scores, k):
    """Indices of `sorted(range(n), key=score, reverse=True)[:k]` — Python's stable sort keeps input
    order on ties (retrieval.py:320-322): tie-break = lower index first."""
    s = np.asarray(scores, dtype=np.float64)
    order = np.lexsort((np.arange(len(s)), -s))
    return order[:k]


# ---------------------------------------------------------------------------------------------
# a2  dense exact scan   (rag/document_store.py:448-460, database/operations.py:126-137:
#      `ORDER BY embedding <=> q LIMIT k`, score = 1 - distance, WHERE agent_id = X)
# ---------------------------------------------------------------------------------------------


def dense_topk(corpus, queries, k, tenant_of_row=None, tenant=None):
    """Exact scan. corpus [N,D] float32, queries [Q,D] -> (ids [Q,k] int64 (-1 padded), scores [Q,k] f64).
    Order: cosine descending, lower row index first on ties; rows of other tenants are skipped."""
    C = np.asarray(corpus)
    Qm = np.asarray(queries)
    ids = np.full((Qm.shape[0], k), -1, dtype=np.int64)
    sc = np.zeros((Qm.shape[0], k), dtype=np.float64)
    keep = np.ones(C.shape[0], dtype=bool) if tenant_of_row is None else (np.asarray(tenant_of_row) == tenant)
    rows = np.nonzero(keep)[0]
    S = cosine_matrix(Qm, C[rows])
    for qi in range(Qm.shape[0]):
        top = stable_topk_desc(S[qi], k)
        ids[qi, :len(top)] = rows[top]
        sc[qi, :len(top)] = S[qi, top]
    return ids, sc


# ---------------------------------------------------------------------------------------------
# a4  keyword scores   (rag/retrieval.py:324-360)
# ---------------------------------------------------------------------------------------------


def tokenize(text):
    """`str.lower().split()` — no stemming, punctuation stays attached (retrieval.py:334-335)."""
    return text.lower().split()


def simple_keyword_scores(query, corpus):
    """|Q ∩ D| / |Q| on lower-cased whitespace token SETS (retrieval.py:349-360)."""
    qt = set(tokenize(query))
    return [(len(qt & set(tokenize(d))) / len(qt)) if qt else 0.0 for d in corpus]


class BM25Okapi:
    """Restatement of rank-bm25 0.2.2 `BM25Okapi` (third-party, absent; see module header).
    k1=1.5, b=0.75, epsilon=0.25; idf = ln(N-df+0.5) - ln(df+0.5); negative idf -> epsilon*mean(idf)
    (mean over the vocabulary in first-appearance order); avgdl = total tokens / N."""

    def __init__(self, tokenized_corpus, k1=1.5, b=0.75, epsilon=0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = len(tokenized_corpus)
        self.doc_len = [len(d) for d in tokenized_corpus]
        self.doc_freqs = []
        nd = {}
        for d in tokenized_corpus:
            f = {}
            for w in d:
                f[w] = f.get(w, 0) + 1
            self.doc_freqs.append(f)
            for w in f:
                nd[w] = nd.get(w, 0) + 1
        self.avgdl = sum(self.doc_len) / self.corpus_size
        self.idf = {}
        idf_sum = 0
        neg = []
        for w, df in nd.items():
            v = math.log(self.corpus_size - df + 0.5) - math.log(df + 0.5)
            self.idf[w] = v
            idf_sum += v
            if v < 0:
                neg.append(w)
        self.average_idf = idf_sum / len(self.idf)
        eps = self.epsilon * self.average_idf
        for w in neg:
            self.idf[w] = eps

    def get_scores(self, query_tokens):
        score = np.zeros(self.corpus_size)
        dl = np.array(self.doc_len)
        for q in query_tokens:                                   # repeats count again
            tf = np.array([(d.get(q) or 0) for d in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (tf * (self.k1 + 1) /
                                               (tf + self.k1 * (1 - self.b + self.b * dl / self.avgdl)))
        return score


def bm25_scores_csr(indptr, doc, tf, doc_len, idf, avgdl, query_terms, k1=1.5, b=0.75):
    """BM25Okapi.get_scores from term-major CSR postings (the index-level form of the same arithmetic, for corpora too
    large to hold as token dicts): for every query term IN ORDER, score[doc] += idf[t] * (tf*(k1+1) / (tf + k1*(1 - b +
    b*dl/avgdl))) over the term's postings. Documents outside a posting list have tf = 0 and receive + 0.0, so the
    sparse update is bit-identical to the dense one above (tests/test_oracle_bm25.py checks that). -1 = unknown term."""
    score = np.zeros(len(doc_len))
    dl = np.asarray(doc_len)
    for t in query_terms:
        if t < 0 or t >= len(idf):
            continue
        a, e = int(indptr[t]), int(indptr[t + 1])
        d = np.asarray(doc[a:e])
        f = np.asarray(tf[a:e])
        score[d] += (idf[t] or 0) * (f * (k1 + 1) / (f + k1 * (1 - b + b * dl[d] / avgdl)))
    return score


def bm25_scores(query, corpus):
    """retrieval.py:324-347: zeros for an empty / all-whitespace corpus; else BM25Okapi scores divided
    by their max when that max is > 0 (else by 1.0)."""
    if not corpus or all(len(d.split()) == 0 for d in corpus):
        return [0.0] * len(corpus)
    bm = BM25Okapi([tokenize(d) for d in corpus])
    s = bm.get_scores(tokenize(query))
    mx = max(s) if len(s) > 0 and max(s) > 0 else 1.0
    return [float(v / mx) for v in s]


# ---------------------------------------------------------------------------------------------
# a3  hybrid_search   (rag/retrieval.py:214-322)
# ---------------------------------------------------------------------------------------------

INTENT_WEIGHTS = {                                               # retrieval.py:22-47
    'question_answering': (0.55, 0.40, 0.05), 'fact_checking': (0.50, 0.45, 0.05),
    'multi_hop_reasoning': (0.60, 0.30, 0.10), 'comparison': (0.50, 0.45, 0.05),
    'summarization': (0.65, 0.25, 0.10), 'search': (0.45, 0.50, 0.05),
    'clarification': (0.70, 0.20, 0.10), 'conversational': (0.70, 0.20, 0.10),
    'default': (0.55, 0.35, 0.10),
}


def weights_for_intent(intent):
    key = intent.lower().replace(' ', '_') if intent else 'default'        # retrieval.py:101
    return INTENT_WEIGHTS.get(key, INTENT_WEIGHTS['default'])


def temporal_scores(n, metadata, now, enable=True, recency_weight=0.15, half_life_days=30):
    """retrieval.py:266-292 with config.py:38-40 defaults; `now` is an explicit naive datetime."""
    if not (metadata and enable):
        return [0.0] * n
    out = []
    for md in metadata:
        ts = md.get('created_at') or md.get('uploaded_at')
        val = 0.0
        if ts:
            if isinstance(ts, str):
                try:
                    ts = datetime.fromisoformat(ts.replace('Z', '+00:00'))
                except ValueError:
                    ts = None
            if ts:
                days_old = (now - ts).total_seconds() / 86400
                val = recency_weight * (0.5 ** (days_old / half_life_days))
        out.append(val)
    return out


def hybrid_search(query, corpus, embeddings, query_embedding, top_k=10, metadata=None, intent=None,
                  default_weights=(0.55, 0.35, 0.10), use_adaptive_weights=True, bm25_available=False,
                  now=None):
    """Returns (idx, rows) where rows[i] = dict(hybrid, semantic, keyword, temporal)."""
    if use_adaptive_weights and intent:
        alpha, beta, gamma = weights_for_intent(intent)
    else:
        alpha, beta, gamma = default_weights
    sem = [cosine(query_embedding, e) for e in embeddings]
    kw = bm25_scores(query, corpus) if bm25_available else simple_keyword_scores(query, corpus)
    tmp = temporal_scores(len(corpus), metadata, now)
    hyb = [alpha * sem[i] + beta * kw[i] + gamma * tmp[i] for i in range(len(corpus))]
    idx = stable_topk_desc(hyb, top_k)
    return [int(i) for i in idx], [dict(hybrid_score=hyb[i], semantic_score=sem[i], keyword_score=kw[i],
                                        temporal_score=tmp[i]) for i in idx]


# ---------------------------------------------------------------------------------------------
# a5  reciprocal rank fusion   (rag/reranker.py:224-271)
# ---------------------------------------------------------------------------------------------


def rrf_fuse(lists, k=60, top_k=10):
    """lists: sequences of hashable keys (the reference keys on the `content` string). rank starts at 1;
    score += 1/(k+rank) in list order; first-seen order kept on ties (stable sort).
    A key repeated inside one list adds once per occurrence.
    Returns (keys, scores, ranks): ranks[i][l] = 1-based rank of keys[i]'s FIRST occurrence in list l (0 = absent)."""
    score, first, ranks = {}, [], {}
    for li, lst in enumerate(lists):
        for rank, key in enumerate(lst, start=1):
            v = 1 / (k + rank)
            if key in score:
                score[key] += v
            else:
                score[key] = v
                first.append(key)
                ranks[key] = [0] * len(lists)
            if ranks[key][li] == 0:
                ranks[key][li] = rank
    order = sorted(range(len(first)), key=lambda i: score[first[i]], reverse=True)[:top_k]
    keys = [first[i] for i in order]
    return keys, [score[x] for x in keys], [ranks[x] for x in keys]


# ---------------------------------------------------------------------------------------------
# a8  MMR   (rag/reranker.py:116-195 class version; rag/nodes/helpers.py:183-260 helper version)
# ---------------------------------------------------------------------------------------------


def mmr_class(query_emb, embs, top_k, lam):
    """λ·rel + (1-λ)·(1-max_sim); first pick has diversity 1.0; max() keeps the FIRST maximal element.
    `embs` are the already-valid embeddings. Returns (positions, mmr_scores)."""
    remaining = list(range(len(embs)))
    sel, sel_scores = [], []
    rel = [cosine(query_emb, e, empty_is_zero=True) for e in embs]
    while len(sel) < top_k and remaining:
        best, best_s = None, None
        for i in remaining:
            if sel:
                div = 1 - max(cosine(embs[i], embs[s], empty_is_zero=True) for s in sel)
            else:
                div = 1.0
            s = lam * rel[i] + (1 - lam) * div
            if best is None or s > best_s:
                best, best_s = i, s
        sel.append(best)
        sel_scores.append(best_s)
        remaining.remove(best)
    return sel, sel_scores


def mmr_helper(query_emb, embs, k, lam):
    """λ·rel − (1-λ)·max_sim; returns all positions unchanged when len <= k (helpers.py:206-207)."""
    n = len(embs)
    if n <= k:
        return list(range(n))
    remaining = list(range(n))
    sel = []
    rel = [cosine(query_emb, e) for e in embs]
    while len(sel) < k and remaining:
        best, best_s = None, None
        for i in remaining:
            ms = max(cosine(embs[i], embs[s]) for s in sel) if sel else 0.0
            s = lam * rel[i] - (1 - lam) * ms
            if best is None or s > best_s:
                best, best_s = i, s
        sel.append(best)
        remaining.remove(best)
    return sel


# ---------------------------------------------------------------------------------------------
# a9  consistency checker   (rag/consistency_checker.py:33-261)
# ---------------------------------------------------------------------------------------------

_META = [r'^(this|that|these|those|it|they)\s+(is|are|was|were)', r'^(here|there)\s+(is|are)',
         r'^(in conclusion|in summary|overall|finally)']
_NEG = [("is not", "is"), ("are not", "are"), ("was not", "was"), ("were not", "were"), ("does not", "does"),
        ("do not", "do"), ("did not", "did"), ("cannot", "can"), ("will not", "will"), ("should not", "should"),
        ("no", "yes"), ("false", "true"), ("incorrect", "correct"), ("never", "always")]


def extract_claims(text):                                         # consistency_checker.py:114-146
    out = []
    for s in re.split(r'[.!?]+', text):
        s = s.strip()
        if len(s) < 20 or any(re.match(p, s.lower()) for p in _META):
            continue
        out.append(s)
    return out


def is_contradiction(t1, t2):                                     # consistency_checker.py:193-239
    a, b = t1.lower(), t2.lower()
    for neg, pos in _NEG:
        if (neg in a and pos in b) or (pos in a and neg in b):
            return True
    n1 = re.findall(r'\b\d+\.?\d*\b', t1)
    n2 = re.findall(r'\b\d+\.?\d*\b', t2)
    return bool(n1 and n2 and set(n1) != set(n2))


def check_consistency(documents, embed_batch, threshold=0.85):    # consistency_checker.py:33-112
    if len(documents) < 2:
        return {"consistent": True, "contradictions": [], "confidence": 1.0, "warning": None}
    claims = []
    for idx, doc in enumerate(documents):
        for c in extract_claims(doc.get("content", "")):
            claims.append({"text": c, "doc_idx": idx, "source": doc.get("source", f"doc_{idx}")})
    if len(claims) < 2:
        return {"consistent": True, "contradictions": [], "confidence": 1.0,
                "warning": "Too few claims to check consistency"}
    try:
        emb = embed_batch([c["text"] for c in claims])
    except Exception:
        emb = None                                                # :162-166 -> no contradictions
    contr = []
    if emb is not None:
        S = cosine_matrix(emb, emb)
        for i in range(len(claims)):
            for j in range(i + 1, len(claims)):
                if claims[i]["doc_idx"] == claims[j]["doc_idx"]:
                    continue
                sim = float(S[i, j])
                if sim >= threshold and is_contradiction(claims[i]["text"], claims[j]["text"]):
                    contr.append({"claim_1": claims[i]["text"][:200], "claim_2": claims[j]["text"][:200],
                                  "source_1": claims[i]["source"], "source_2": claims[j]["source"],
                                  "similarity": round(sim, 3), "type": "semantic_contradiction"})
    total_pairs = len(claims) * (len(claims) - 1) / 2
    score = 1.0 - min(len(contr) / max(total_pairs, 1), 1.0)
    n = len(contr)
    warning = None
    if n == 1:
        warning = "Warning: Found 1 potential contradiction in sources. Response may be unreliable."
    elif 1 < n <= 3:
        warning = f"Warning: Found {n} contradictions in sources. Please verify information."
    elif n > 3:
        warning = f"Warning: Found {n} contradictions in sources. High uncertainty in response."
    return {"consistent": n == 0 or score >= 0.8, "contradictions": contr[:5], "contradiction_count": n,
            "confidence": score, "total_claims": len(claims), "warning": warning}


# ---------------------------------------------------------------------------------------------
# a10 context compressor   (rag/context_compressor.py:56-286)
# ---------------------------------------------------------------------------------------------

_STOP = {'the', 'a', 'an', 'and', 'or', 'but', 'in', 'on', 'at', 'to', 'for', 'of', 'with', 'by', 'from', 'is',
         'was', 'are', 'were', 'be', 'been', 'being'}


def split_sentences(text):                                        # context_compressor.py:207-215
    if not text:
        return []
    return [s.strip() for s in re.split(r'[.!?]+\s+', text) if len(s.strip()) > 20]


def score_sentence_lexical(query, sentence):                      # context_compressor.py:265-286
    ql, sl = query.lower(), sentence.lower()
    qw = set(re.findall(r'\b\w+\b', ql)) - _STOP
    sw = set(re.findall(r'\b\w+\b', sl)) - _STOP
    if not qw:
        return 0.0
    score = len(qw & sw) / len(qw)
    if ql in sl:
        score += 0.2
    return min(score, 1.0)


def score_sentences_hybrid(query, sentences, q_emb, s_embs):      # context_compressor.py:217-241
    return [0.7 * cosine(q_emb, e) + 0.3 * score_sentence_lexical(query, s) for s, e in zip(sentences, s_embs)]


# ---------------------------------------------------------------------------------------------
# a6/a7 re-ranker post-processing   (rag/reranker.py:28-90, 320-384)
# ---------------------------------------------------------------------------------------------


def longest_first_lengths(n1, n2, max_content):
    """Lengths the two sequences of a pair keep under truncation='longest_first' (what CrossEncoder.predict's tokenizer
    call applies, /root/reference/rag/reranker.py:355 -> sentence-transformers -> fast tokenizer; third-party, absent from
    the reference tree): max_content = max_length - special tokens. The shorter side is kept whole while it fits; if both
    exceed their share the FIRST-or-shorter side gets floor(max/2), the other the rest. Pinned against the `tokenizers`
    package shipped in this image by tests/test_pair_truncation.py."""
    if n1 + n2 <= max_content:
        return n1, n2
    swap = n1 > n2
    a, b = (n2, n1) if swap else (n1, n2)
    b = a if a > max_content else max(a, max_content - a)
    if a + b > max_content:
        a = max_content // 2
        b = a + max_content % 2
    return (b, a) if swap else (a, b)


def sigmoid(x):
    return 1 / (1 + math.exp(-x))                                 # reranker.py:359


def openai_rerank_scores(query_emb, content_embs, originals):     # reranker.py:67-77
    return [0.7 * cosine(query_emb, e) + 0.3 * o for e, o in zip(content_embs, originals)]
