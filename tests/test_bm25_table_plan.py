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
