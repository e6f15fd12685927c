"""GPU: BM25 through the two-level bracket tables (csrc/bm25.hip) against the CSR oracle, bit for bit. Replaces
BM25Okapi.get_scores + /max of /root/reference/rag/retrieval.py:324-347. Every term class of the table plan is hit: direct
per-range rows (long lists), coarse rows with an in-bracket search (middle of the distribution), no row at all (rare terms),
lists that sit inside ONE bracket, postings exactly on bracket / range edges, the last partial range, empty terms."""
import numpy as np
import pytest

from oracle import rag_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from optimized_rag_amd import RagEngine
    e = RagEngine(dim=64, device=0)
    yield e
    e.close()


def _postings(n_docs, doc_lists, rng):
    from optimized_rag_amd.bm25 import Bm25Postings
    docs = [np.unique(np.asarray(d, dtype=np.int64)).astype(np.int32) for d in doc_lists]
    indptr = np.concatenate([[0], np.cumsum([d.shape[0] for d in docs])]).astype(np.int64)
    doc = np.concatenate(docs) if docs else np.zeros(0, np.int32)
    tf = rng.integers(1, 5, doc.shape[0]).astype(np.int32)
    doc_len = rng.integers(5, 60, n_docs).astype(np.int32)
    idf = Bm25Postings.idf_table(np.diff(indptr), n_docs)
    return Bm25Postings(indptr, doc, tf, doc_len, idf, float(doc_len.sum()) / n_docs)


def _check(eng, post, terms_of, k):
    ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
    terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
    ids, rows, scores, mx = eng.bm25_topk(ptr, terms, k)
    eng.set_option("bm25_no_staging", 1)
    try:
        _, rows_x, scores_x, _ = eng.bm25_topk(ptr, terms, k)
    finally:
        eng.set_option("bm25_no_staging", 0)
    np.testing.assert_array_equal(rows, rows_x)
    np.testing.assert_array_equal(scores, scores_x)
    for qi, t in enumerate(terms_of):
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, t)
        m = raw.max() if raw.max() > 0 else 1.0
        top = O.stable_topk_desc(raw, k)
        np.testing.assert_array_equal(rows[qi], top.astype(np.int32), err_msg=str(t))
        np.testing.assert_array_equal(scores[qi], raw[top] / m, err_msg=str(t))
        assert mx[qi] == m


def test_every_table_class_against_the_csr_oracle(eng):
    rng = np.random.default_rng(2048)
    n_docs = 700_000                                             # 342 ranges of 2048 documents, the last one partial
    lists = [
        rng.integers(0, n_docs, 260_000),                        # 0  long list: direct per-range row
        rng.choice(n_docs, 1372, replace=False),                 # 1  exactly 4 x 343 postings: the shortest list with a direct row
        rng.choice(n_docs, 1371, replace=False),                 # 2  one posting fewer: 4096-document brackets
        rng.choice(n_docs, 300, replace=False),                  # 3  16384-document brackets
        5 * 16384 + rng.choice(16384, 300, replace=False),       # 4  the whole list inside ONE bracket (long in-bracket search)
        np.arange(0, n_docs, 16384),                             # 5  postings exactly on bracket edges
        np.arange(2048, n_docs, 2048 * 7),                       # 6  ... and on range edges
        rng.choice(n_docs, 7, replace=False),                    # 7  7 postings: no table
        rng.choice(n_docs, 12, replace=False),                   # 8  12 postings: three entries of 2^19 documents
        699_000 + rng.choice(1000, 400, replace=False),          # 9  only in the last (partial) range and its neighbour
        np.arange(n_docs),                                       # 10 every document
        [],                                                      # 11 a term of the vocabulary without postings
        [n_docs - 1],                                            # 12 the very last document
        [0],                                                     # 13 the very first one
    ]
    for _ in range(30):                                          # + document frequencies log-uniform in [1, 50k]
        lists.append(rng.choice(n_docs, int(np.exp(rng.uniform(0, np.log(50_000)))), replace=False))
    post = _postings(n_docs, lists, rng).load(eng)
    queries = [[0, 3, 7], [1, 2], [4], [4, 5, 6], [7, 8], [9, 12, 13], [10, 3], [11], [11, 4, -1], [12], [13], [8, 8, 8, 2],
               [5], [6, 9], [14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25], [30, 31, 32, 33], [40, 41, 42, 43, 0], [2, 4, 6, 8]]
    for k in (20, 100):
        _check(eng, post, queries, k)


def test_all_document_scores_through_the_tables(eng):
    """mode 1 of the range kernel (rag_bm25_scores_host: BM25Okapi.get_scores for every document, the input of hybrid_search's
    linear fusion) on an ad-hoc sized corpus: 9,000 documents (5 ranges), every table class again, raw float64 bit-exact."""
    rng = np.random.default_rng(9)
    n_docs = 9_000
    lists = [rng.choice(n_docs, n, replace=False) for n in (9000, 4000, 24, 23, 9, 8, 7, 1)] + [[], np.arange(2048, 4096), [8999]]
    post = _postings(n_docs, lists, rng).load(eng)
    terms_of = [[0, 1, 2], [3, 4, 5, 6, 7], [8], [9, 10], [2, 2, 10, -1, 7]]
    ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
    terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
    got = eng.bm25_scores(ptr, terms)
    adhoc = eng.bm25_scores_adhoc(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, ptr, terms)
    for qi, t in enumerate(terms_of):
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, t)
        np.testing.assert_array_equal(got[qi], raw)
        np.testing.assert_array_equal(adhoc[qi], raw)


def test_two_million_term_zipf_vocabulary(eng):
    """The vocabulary size that broke the dense table: 1M documents, 2,000,000 term ids with a truncated
    Zipf(1.1) tail (bench_modes.zipf_postings_gpu): the table is a small fraction of the postings, and batch queries drawn
    from documents (mostly frequent terms, some rare) agree with the CSR oracle on sampled queries and, for the whole
    batch, with the exhaustive per-range select."""
    import torch
    import bench_modes as BM
    from optimized_rag_amd._lib import bm25_index_bytes
    from optimized_rag_amd.bm25 import Bm25Postings
    N, V, Q, k = 1_000_000, 2_000_000, 256, 100
    qdocs = np.random.default_rng(3).integers(0, N, Q)
    indptr, d, tf, dl, sampled = BM.zipf_postings_gpu(N, V, 120, torch.device("cuda", 0), sample_docs=qdocs)
    df = np.diff(indptr)
    assert (df > 0).sum() > 1_500_000                             # the tail is populated: >= 1.5M distinct terms occur
    post_b, meta_b, tab_b = bm25_index_bytes(indptr, N)
    assert tab_b <= indptr[-1] and tab_b < 0.09 * post_b           # dense table: 2M x 490 x 4 B = 3.9 GB
    post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(df, N), float(dl.sum()) / N)
    post.idf[df == 0] = 0.0
    post.load(eng)
    ptr, terms = BM.term_queries_from_docs(sampled, qdocs)
    rare = np.nonzero((df > 0) & (df < 8))[0][:3]                  # make sure table-less terms are queried too
    terms[ptr[5]:ptr[5] + 3] = rare
    ptr_d, terms_d = torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda()

    def run():
        ids = torch.empty((Q, k), dtype=torch.int64, device="cuda")
        rows = torch.empty((Q, k), dtype=torch.int32, device="cuda")
        sc = torch.empty((Q, k), dtype=torch.float64, device="cuda")
        eng.bm25_topk_dev(ptr_d, terms_d, k, ids, rows, sc)
        torch.cuda.synchronize()
        return rows.cpu().numpy(), sc.cpu().numpy()

    rows, sc = run()
    eng.set_option("bm25_no_staging", 1)
    try:
        rows_x, sc_x = run()
    finally:
        eng.set_option("bm25_no_staging", 0)
    np.testing.assert_array_equal(rows, rows_x)
    np.testing.assert_array_equal(sc, sc_x)
    for qi in (0, 5, 100, 255):
        raw = O.bm25_scores_csr(indptr, d, tf, dl, post.idf, post.avgdl, terms[ptr[qi]:ptr[qi + 1]].tolist())
        top = O.stable_topk_desc(raw, k)
        np.testing.assert_array_equal(rows[qi], top.astype(np.int32))
        np.testing.assert_array_equal(sc[qi], raw[top] / (raw.max() if raw.max() > 0 else 1.0))


def test_packed_postings_equal_the_twelve_byte_form(eng):
    """Option bm25_packed: the scoring loop streams 4-byte postings (document number inside its range | code of its (tf, doc length) pair) and
    multiplies idf by a table value g[code] instead of streaming (doc i32, impact f64). Both are the same float64 product
    (`idf * (num / den)`, rank-bm25's association), so top-k rows, scores and all-document scores must be bit-identical between
    an index loaded with the option and a default one - and equal to the CSR oracle. Term frequencies up to 300
    and document lengths spread over thousands of values exercise a large code table."""
    from optimized_rag_amd import RagEngine
    rng = np.random.default_rng(77)
    n_docs = 40_000
    lists = [rng.choice(n_docs, n, replace=False) for n in (40_000, 21_000, 5000, 700, 64, 9, 3)]
    post = _postings(n_docs, lists, rng)
    post.tf[:] = rng.integers(1, 301, post.tf.shape[0]).astype(np.int32)
    post.doc_len[:] = rng.integers(1, 6000, n_docs).astype(np.int32)
    post.avgdl = float(post.doc_len.sum()) / n_docs
    terms_of = [[0, 1, 2], [3, 4, 5, 6], [0], [2, 2, 6, -1], [1, 3, 5]]
    ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
    terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
    out = {}
    for name, flag in (("packed", 1), ("wide", 0)):
        e = RagEngine(dim=64, device=0)
        try:
            e.set_option("bm25_packed", flag)
            post.load(e)
            out[name] = (e.bm25_topk(ptr, terms, 50), e.bm25_scores(ptr, terms))
        finally:
            e.close()
    for a, b in zip(out["packed"][0], out["wide"][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out["packed"][1], out["wide"][1])
    for qi, t in enumerate(terms_of):
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, t)
        np.testing.assert_array_equal(out["packed"][1][qi], raw)


def test_column_workgroup_order_equals_range_major(eng):
    """From 128 queries on, the scoring workgroups are ordered in XCD-aware columns (csrc/bm25.hip bm_make_grid: every query of a
    column on the same 2048-document range, the queries of a range split into up to 8 columns when a launch has few ranges, empty
    workgroups where columns x queries do not fill the grid). Placement must not change a bit: top-k rows / scores / maxima and
    all-document scores equal the range-major order (option bm25_linear_grid) and the CSR oracle, for query counts that hit
    every split (129: groups of 65 with padding; 200; 260: ragged last group) on 3 and on 45 ranges."""
    rng = np.random.default_rng(4242)
    for n_docs in (5_000, 91_000):
        lists = [rng.choice(n_docs, int(np.exp(rng.uniform(0, np.log(n_docs)))), replace=False) for _ in range(60)]
        lists[0] = np.arange(n_docs)
        post = _postings(n_docs, lists, rng).load(eng)
        for Q in (129, 200, 260):
            terms_of = [list(rng.integers(-1, 60, int(rng.integers(1, 9)))) for _ in range(Q)]
            ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
            terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
            got = eng.bm25_topk(ptr, terms, 30)
            dense = eng.bm25_scores(ptr, terms) if n_docs == 5_000 else None
            eng.set_option("bm25_linear_grid", 1)
            try:
                ref = eng.bm25_topk(ptr, terms, 30)
                dense_ref = eng.bm25_scores(ptr, terms) if n_docs == 5_000 else None
            finally:
                eng.set_option("bm25_linear_grid", 0)
            for a, b in zip(got, ref):
                np.testing.assert_array_equal(a, b)
            if dense is not None:
                np.testing.assert_array_equal(dense, dense_ref)
            for qi in (0, 64, 65, 128, Q - 1):
                raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, [int(x) for x in terms_of[qi]])
                top = O.stable_topk_desc(raw, 30)
                np.testing.assert_array_equal(got[1][qi], top.astype(np.int32))
                np.testing.assert_array_equal(got[2][qi], raw[top] / (raw.max() if raw.max() > 0 else 1.0))
                if dense is not None:
                    np.testing.assert_array_equal(dense[qi], raw)


def test_selection_merge_equals_the_sorting_merge(eng):
    """The partial lists of a stage are folded into the running top-k by a per-wave bitwise SELECTION (csrc/bm25.hip
    bm25_merge_select_kernel; k <= 256) instead of a bitonic sort: same list bit for bit, including tie plateaus cut by the lowest
    rows. Single-term queries over tf = 1 and one document length make every score of a list EQUAL (a plateau across all ranges:
    the row-order cut decides everything); mixed queries give ordinary lists; k = 100 / 256 / 7 / 300; 45 ranges (three stages)."""
    from optimized_rag_amd.bm25 import Bm25Postings
    rng = np.random.default_rng(808)
    n_docs = 91_000
    lists = [rng.choice(n_docs, n, replace=False) for n in (91_000, 60_000, 30_000, 5_000, 800, 90, 9)]
    docs = [np.sort(np.asarray(d, dtype=np.int64)).astype(np.int32) for d in lists]
    indptr = np.concatenate([[0], np.cumsum([d.shape[0] for d in docs])]).astype(np.int64)
    doc = np.concatenate(docs)
    tf = np.ones(doc.shape[0], np.int32)
    tf[indptr[2]:indptr[3]] = rng.integers(1, 3, docs[2].shape[0])                 # term 2: two score levels, two plateaus
    doc_len = np.full(n_docs, 40, np.int32)
    post = Bm25Postings(indptr, doc, tf, doc_len, Bm25Postings.idf_table(np.diff(indptr), n_docs), 40.0).load(eng)
    terms_of = [[0], [1], [2], [3], [4], [5], [6], [0, 1], [2, 3], [1, 2, 3, 4], [6, 5], [4, 4]] + \
               [list(rng.integers(0, 7, int(rng.integers(1, 5)))) for _ in range(130)]
    ptr = np.cumsum([0] + [len(t) for t in terms_of]).astype(np.int32)
    terms = np.asarray([x for t in terms_of for x in t], dtype=np.int32)
    for k in (100, 256, 7, 300):                                               # 300 > 256: the sorting kernel is the only path
        got = eng.bm25_topk(ptr, terms, k)
        eng.set_option("bm25_sort_merge", 1)
        try:
            ref = eng.bm25_topk(ptr, terms, k)
        finally:
            eng.set_option("bm25_sort_merge", 0)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        for qi in (0, 2, 5, 6, 9, 11, 141):
            raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, [int(x) for x in terms_of[qi]])
            top = O.stable_topk_desc(raw, k)
            live = raw[top] > 0 if raw.max() > 0 else np.ones(k, bool)
            np.testing.assert_array_equal(got[1][qi][live], top.astype(np.int32)[live], err_msg=f"k {k} query {qi}")
            np.testing.assert_array_equal(got[2][qi], raw[top] / (raw.max() if raw.max() > 0 else 1.0))
