"""CPU: the host-side plan of the BM25 bracket tables (rag_bm25_index_bytes; no GPU call). A term's table is sized by its
document frequency, so the index survives the vocabulary the reference tokeniser produces (`doc.lower().split()`,
/root/reference/rag/retrieval.py:334-335: millions of distinct terms on a large shard) - round 2's dense V x n_ranges table
was 122 GB for 5M terms on a 12.5M-document shard."""
import numpy as np

from optimized_rag_amd._lib import bm25_index_bytes

RANGE = 2048


def _indptr(df):
    return np.concatenate([[0], np.cumsum(np.asarray(df, dtype=np.int64))])


def _entries(n_docs, df):
    """Restatement of the plan (csrc/bm25.hip bm_plan_term): smallest bracket width 2^g >= 2048 docs whose table
    (ceil(n_pad / 2^g) + 1 entries) fits df / 4 entries; no table when even two brackets do not fit."""
    n_pad = -(-n_docs // RANGE) * RANGE
    g = 11
    ent = lambda gg: -(-n_pad // (1 << gg)) + 1
    while ent(g) > df // 4 and ent(g) > 2:
        g += 1
    return 0 if (ent(g) > df // 4 or ent(g) <= 2) else ent(g)


def test_plan_matches_restatement_per_term_class():
    n_docs = 700_000                                  # 342 ranges -> a direct row has 343 entries
    cases = {0: 0, 1: 0, 7: 0, 8: 0, 11: 0, 12: 3, 300: 44, 1371: 172, 1372: 343, 200_000: 343, n_docs: 343}
    for df, want in cases.items():
        assert _entries(n_docs, df) == want, df
        post, meta, tab = bm25_index_bytes(_indptr([df]), n_docs)
        assert tab == 4 * want, (df, tab)
        assert post == (df + 8) * 12 and meta == 32


def test_table_is_bounded_by_the_postings_for_any_vocabulary():
    """5M terms with Zipf document frequencies on a 12.5M-document shard: the table stays under nnz bytes (1/12 of the
    postings) where the dense form needs V x (n_ranges + 1) x 4 B = 122 GB."""
    n_docs, V = 12_500_000, 5_000_000
    r = np.arange(1, V + 1, dtype=np.float64)
    df = np.minimum(n_docs, np.maximum(1, (1.2e9 * r ** -1.1 / (r ** -1.1).sum()))).astype(np.int64)
    post, meta, tab = bm25_index_bytes(_indptr(df), n_docs)
    nnz = int(df.sum())
    assert post == (nnz + 8) * 12 and meta == 32 * V
    assert tab <= nnz                                                # 4 B x (<= df / 4 entries) per term
    dense = V * (-(-n_docs // RANGE) + 1) * 4
    assert dense > 100e9 and tab < 1.3e9, (dense, tab)
    # the bench vocabulary (100k folded-Zipf ids at 1M docs): long lists keep their direct rows, the table is no larger than r2's
    df_b = np.maximum(1, (9.5e7 * r[:100_000] ** -0.6 / (r[:100_000] ** -0.6).sum())).astype(np.int64)
    _, _, tab_b = bm25_index_bytes(_indptr(df_b), 1_000_000)
    assert tab_b <= 100_000 * 490 * 4


def test_empty_and_degenerate_inputs():
    assert bm25_index_bytes(_indptr([]), 10) == (96, 0, 0)
    assert bm25_index_bytes(_indptr([0, 0, 5]), 10) == ((5 + 8) * 12, 96, 0)


def _decode(plan, b):
    """The scoring kernel's workgroup -> (range, query) rule (csrc/bm25.hip bm25_range_kernel), restated."""
    blocks, nr, nq, G, L = plan
    if L == 0:
        return b % nr, b // nr
    x, s = b % 8, b // 8
    col, qi = x + 8 * (s // L), s % L
    rl, q = col // G, (col % G) * L + qi
    return (rl, q) if rl < nr and q < nq else None


def test_every_range_query_pair_is_scored_exactly_once():
    """rag_bm25_grid_plan (host side of the XCD-aware workgroup order): for every launch shape the workgroups cover each (range,
    query) pair once and only once, the padding workgroups decode to nothing, and a column's queries share one range."""
    from optimized_rag_amd._lib import bm25_grid_plan
    rng = np.random.default_rng(12)
    shapes = [(4, 1024), (28, 1024), (224, 1024), (233, 1024), (4, 128), (1, 129), (3, 200), (489, 256), (7, 1000), (28, 257), (1, 1), (5, 127),
              (6105, 256), (9, 2048)] + [(int(rng.integers(1, 400)), int(rng.integers(1, 1500))) for _ in range(12)]
    for nr, nq in shapes:
        plan = bm25_grid_plan(nr, nq)
        blocks, nr_o, nq_o, G, L = plan
        assert (nr_o, nq_o) == (nr, nq) and blocks >= nr * nq
        assert (L == 0) == (nq < 128) and blocks < 4.1 * nr * nq + 64
        if nr * nq > 300_000:                               # large shapes: sampled workgroups + the counting argument
            ids = rng.integers(0, blocks, 20_000)
            seen = {_decode(plan, int(b)) for b in ids} - {None}
            assert all(0 <= r < nr and 0 <= q < nq for r, q in seen)
            assert G * L >= nq and blocks == 8 * -(-nr * G // 8) * L if L else blocks == nr * nq
            continue
        seen = {}
        for b in range(blocks):
            d = _decode(plan, b)
            if d is not None:
                assert d not in seen, (nr, nq, b, d)
                seen[d] = b
        assert len(seen) == nr * nq, (nr, nq, len(seen))
        if L:                                               # the queries of a column sit on ONE XCD (workgroup id % 8) and one range
            by_col = {}
            for (r, q), b in seen.items():
                by_col.setdefault((r, q // L), set()).add(b % 8)
            assert all(len(v) == 1 for v in by_col.values())
        lin = bm25_grid_plan(nr, nq, linear=True)
        assert lin[4] == 0 and lin[0] == nr * nq
