"""GPU parity: dense cosine top-k through the C-ABI vs the CPU oracle (bit-exact ids, scores within 1e-9;
the north-star tolerance is 1e-3 — float64 rescoring makes it far tighter)."""
import os

import numpy as np
import pytest

from oracle import rag_oracle as O

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-9          # float64 on both sides, different summation order


@pytest.fixture(scope="module")
def eng_factory():
    from optimized_rag_amd import RagEngine
    made = []

    def make(dim):
        e = RagEngine(dim=dim, device=0)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()


def planted_queries(rng, corpus, Q, noise=0.5):
    r = rng.integers(0, corpus.shape[0], Q)
    return (corpus[r] + noise * rng.standard_normal((Q, corpus.shape[1]))).astype(np.float32)


def check(eng, corpus, queries, k, tenant_of_row=None, tenant=-1, ids=None):
    got_ids, got_rows, got_sc = eng.dense_topk(queries, k, tenant=tenant)
    oid, osc = O.dense_topk(corpus, queries, k, tenant_of_row, tenant if tenant >= 0 else None)
    np.testing.assert_array_equal(got_rows, oid.astype(np.int32))
    exp_ids = oid if ids is None else np.where(oid >= 0, ids[np.maximum(oid, 0)], -1)
    np.testing.assert_array_equal(got_ids, exp_ids)
    np.testing.assert_allclose(got_sc, osc, rtol=0, atol=SCORE_TOL)
    st = eng.dense_stats()
    assert st["proven_fast"] + st["proven_wide"] + st["exact_scan"] == queries.shape[0], st
    return st


@pytest.mark.parametrize("N,D,Q,k", [
    (300, 1536, 5, 20),          # single dense stage, N not a tile multiple
    (2048, 1536, 16, 20),        # exactly stage 0
    (5000, 1536, 33, 20),        # two stages
    (40000, 1536, 64, 20),       # three stages
    (40000, 64, 300, 5),         # small dim (padded K = 64), Q not a tile multiple
    (9000, 384, 7, 100),         # large k -> larger shortlist
    (70000, 128, 3, 1),
    (17, 1536, 4, 20),           # fewer rows than k -> -1 padding
    (2080, 64, 3, 1),            # 9 tiles: the one-tile trailing stage is merged into the dense stage 0 (2304 slots, not 2048)
    (2304, 128, 5, 20),
    (2305, 64, 4, 7),            # 10 tiles: the trailing stage stands on its own
])
def test_dense_topk_matches_oracle(eng_factory, N, D, Q, k):
    rng = np.random.default_rng(N + D + Q + k)
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    queries = planted_queries(rng, corpus, Q)
    eng = eng_factory(D)
    eng.index_load(corpus)
    st = check(eng, corpus, queries, k)
    assert st["overflowed"] == 0


def test_unnormalised_rows_and_scale_invariance(eng_factory):
    rng = np.random.default_rng(5)
    N, D = 6000, 1536
    corpus = (rng.standard_normal((N, D)) * rng.uniform(1e-3, 1e3, (N, 1))).astype(np.float32)
    queries = (planted_queries(rng, corpus, 12) * 37.5).astype(np.float32)
    eng = eng_factory(D)
    eng.index_load(corpus)
    check(eng, corpus, queries, 20)


def test_duplicates_ties_and_zero_rows(eng_factory):
    """Exact duplicate rows tie; order must be lower row first (Python's stable sort). Zero rows score 0.0."""
    rng = np.random.default_rng(6)
    N, D = 8000, 256
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    base = corpus[10].copy()
    dup_rows = rng.choice(np.arange(100, N), 300, replace=False)
    corpus[dup_rows] = base                       # 301 identical rows (incl. row 10): more survivors than the fast path ranks
    corpus[20:30] = 0.0                           # zero-norm rows
    corpus[31, 5] = np.inf                        # non-finite row -> scores 0.0, never NaN
    queries = np.stack([base + 0.3 * rng.standard_normal(D), base, rng.standard_normal(D), np.zeros(D)]).astype(np.float32)
    eng = eng_factory(D)
    eng.index_load(corpus)
    oc = corpus.copy()
    oc[31] = 0.0                                  # oracle convention for a non-finite row: cosine 0.0
    got_ids, got_rows, got_sc = eng.dense_topk(queries, 20)
    oid, osc = O.dense_topk(oc, queries, 20)
    np.testing.assert_array_equal(got_rows, oid.astype(np.int32))
    np.testing.assert_allclose(got_sc, osc, atol=SCORE_TOL)
    assert (got_sc[3] == 0.0).all() and (got_rows[3] == np.arange(20)).all()     # zero query: all ties at 0.0
    st = eng.dense_stats()
    assert st["proven_wide"] >= 1 and st["exact_scan"] >= 1     # dup cluster -> wide path; zero query -> buffer overflow -> scan


def test_tenant_filter_and_id_table(eng_factory):
    rng = np.random.default_rng(7)
    N, D = 12000, 512
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    tenants = rng.integers(0, 3, N).astype(np.int32)
    ids = (rng.permutation(N) + 1_000_000_000_000).astype(np.int64)
    queries = planted_queries(rng, corpus, 9)
    eng = eng_factory(D)
    eng.index_load(corpus, ids=ids)
    eng.set_tenants(tenants)
    for t in (0, 2):
        check(eng, corpus, queries, 20, tenant_of_row=tenants, tenant=t, ids=ids)
    check(eng, corpus, queries, 20, ids=ids)                   # no filter
    np.testing.assert_array_equal(eng.fetch_rows([5, 0, N - 1]), corpus[[5, 0, N - 1]])


@pytest.mark.parametrize("level", ["1", "2"])
def test_forced_fallback_levels_give_identical_results(eng_factory, level, monkeypatch):
    rng = np.random.default_rng(8)
    N, D = 20000, 1536
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    queries = planted_queries(rng, corpus, 6)
    eng = eng_factory(D)
    eng.index_load(corpus)
    eng.set_option("force_level", int(level))
    st = check(eng, corpus, queries, 20)
    assert st["proven_fast"] == 0
    if level == "2":
        assert st["exact_scan"] == 6


def test_clustered_corpus(eng_factory):
    """Tight clusters: many near-ties around the k-th score -> exercises the proof / fallback logic."""
    rng = np.random.default_rng(9)
    N, D = 30000, 1536
    centers = rng.standard_normal((20, D)).astype(np.float32)
    corpus = (centers[rng.integers(0, 20, N)] + 0.01 * rng.standard_normal((N, D))).astype(np.float32)
    queries = (centers[:8] + 0.01 * rng.standard_normal((8, D))).astype(np.float32)
    eng = eng_factory(D)
    eng.index_load(corpus)
    check(eng, corpus, queries, 20)


def test_golden_cosine_through_pairwise_kernel(eng_factory, golden_dir):
    g = np.load(os.path.join(golden_dir, "cosine.npz"))
    eng = eng_factory(1536)
    M = eng.pairwise_cosine(g["a"], g["b"])
    np.testing.assert_allclose(np.diag(M), g["exp_retrieval"], atol=1e-12)
    assert M[5, 5] == 0.0 and M[6, 6] == 0.0                    # zero-norm -> exactly 0.0


def test_merge_topk_equals_unsharded_search(eng_factory):
    """Shard the corpus row-wise in 3, search each shard, merge on device == search of the whole corpus."""
    import torch
    rng = np.random.default_rng(10)
    N, D, Q, k = 30000, 256, 40, 20
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    corpus[2000] = corpus[25000]                                 # cross-shard exact tie
    queries = planted_queries(rng, corpus, Q)
    queries[0] = corpus[25000]
    bounds = [0, 9000, 21000, N]
    parts_i, parts_s = [], []
    for s in range(3):
        e = eng_factory(D)
        e.index_load(corpus[bounds[s]:bounds[s + 1]], id_base=bounds[s])
        i, _, sc = e.dense_topk(queries, k)
        parts_i.append(i)
        parts_s.append(sc)
    e = eng_factory(D)
    ids = torch.from_numpy(np.stack(parts_i)).cuda()
    sc = torch.from_numpy(np.stack(parts_s)).cuda()
    oi = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    os_ = torch.empty((Q, k), dtype=torch.float64, device="cuda")
    e.merge_topk_dev(ids, sc, oi, os_)
    torch.cuda.synchronize()
    oid, osc = O.dense_topk(corpus, queries, k)
    np.testing.assert_array_equal(oi.cpu().numpy(), oid)
    np.testing.assert_allclose(os_.cpu().numpy(), osc, atol=SCORE_TOL)


def test_chunked_reserve_append_equals_one_shot_load(eng_factory):
    """rag_index_reserve + rag_index_append_* (host and device blocks) == rag_index_load; searchable while filling."""
    import torch
    rng = np.random.default_rng(12)
    N, D = 9000, 384
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    queries = planted_queries(rng, corpus, 10)
    eng = eng_factory(D)
    eng.index_reserve(N, id_base=500)
    eng.index_append(corpus[:2500])                                       # host block
    ids, rows, sc = eng.dense_topk(queries, 20)                           # search over the rows appended so far
    oid, osc = O.dense_topk(corpus[:2500], queries, 20)
    np.testing.assert_array_equal(rows, oid.astype(np.int32))
    eng.index_append(torch.from_numpy(corpus[2500:7000]).cuda())          # device block
    eng.index_append(corpus[7000:])
    ids, rows, sc = eng.dense_topk(queries, 20)
    oid, osc = O.dense_topk(corpus, queries, 20)
    np.testing.assert_array_equal(rows, oid.astype(np.int32))
    np.testing.assert_array_equal(ids, oid + 500)
    np.testing.assert_allclose(sc, osc, atol=SCORE_TOL)
    with pytest.raises(Exception):
        eng.index_append(corpus[:1])                                      # beyond the reservation: loud error


def test_device_entry_points_are_ordered_with_the_callers_stream(eng_factory):
    """Regression: *_dev calls run on the stream handed in (torch's current stream, which may be the default stream 0).
    A source buffer that is overwritten right after rag_index_append_dev, and outputs consumed right after
    rag_dense_topk_dev without any explicit synchronisation, must still see correctly ordered data."""
    import torch
    D, blk_rows, n_blk = 256, 20000, 12
    eng = eng_factory(D)
    eng.index_reserve(blk_rows * n_blk)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    buf = torch.empty((blk_rows, D), device="cuda")
    firsts = []
    for b in range(n_blk):
        buf.normal_(generator=g)
        firsts.append(buf[:3].clone())
        eng.lib.rag_index_append_dev(eng.h, buf.data_ptr(), blk_rows, torch.cuda.current_stream().cuda_stream)
        eng.n_rows += blk_rows
        buf.fill_(float(b))                                          # clobber the source immediately
    want = torch.cat(firsts).cpu().numpy()
    rows = np.concatenate([np.arange(b * blk_rows, b * blk_rows + 3) for b in range(n_blk)])
    np.testing.assert_array_equal(eng.fetch_rows(rows), want)
    q = torch.from_numpy(want[::3].copy()).cuda()                    # one exact copy per block
    ids = torch.empty((n_blk, 4), dtype=torch.int64, device="cuda")
    sc = torch.empty((n_blk, 4), dtype=torch.float64, device="cuda")
    eng.dense_topk_dev(q, 4, ids, None, sc)
    got = ids[:, 0].clone()                                          # consumer on the same stream, no sync in between
    assert got.cpu().tolist() == [b * blk_rows for b in range(n_blk)]


def test_shard_directory_streams_into_the_index(eng_factory, tmp_path):
    """export rows shaped like document_chunks -> shard directory -> chunked load (whole shard and a
    [begin, end) row range as one rank of a multi-GPU run would) -> search returns the table's primary keys, honours
    the agent_id filter and matches the oracle."""
    from optimized_rag_amd import shard_format as SF
    rng = np.random.default_rng(21)
    N, D = 3000, 384
    emb = rng.standard_normal((N, D)).astype(np.float32)
    pk = (rng.permutation(N) + 50_000).astype(np.int64)
    agents = rng.integers(0, 3, N)
    rows = [(int(pk[i]), f"agent-{agents[i]}", f"doc {i} w{i % 17}", SF.format_pgvector_text(emb[i]) if i % 5 == 0 else emb[i],
             None, None) for i in range(N)]
    sh = SF.open_shard(SF.export_table(rows, str(tmp_path / "s"), dim=D))
    queries = planted_queries(rng, emb, 9)
    eng = eng_factory(D)
    post = SF.load_shard_into(eng, sh, chunk_rows=700)
    assert post is not None and eng.n_rows == N
    check(eng, emb, queries, 20, ids=pk)
    t = sh.tenant_table["agent-1"]
    check(eng, emb, queries, 20, tenant_of_row=np.asarray(sh.tenants), tenant=t, ids=pk)
    b, e = 1000, 2200                                          # one rank's row range
    eng2 = eng_factory(D)
    SF.load_shard_into(eng2, sh, begin=b, end=e, chunk_rows=512)
    check(eng2, emb[b:e], queries, 20, ids=pk[b:e])
    sh.close()


# ---------------------------------------------------------------------------------------- row-order independence (r2)
def _clustered(rng, n_clusters, per, D):
    centers = rng.standard_normal((n_clusters, D))
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    rows = np.repeat(centers, per, axis=0) + rng.standard_normal((n_clusters * per, D)) / np.sqrt(D)
    return rows.astype(np.float32)


@pytest.mark.parametrize("k", [20, 100])
def test_cluster_by_cluster_storage_needs_no_fallback(eng_factory, k):
    """A table stored cluster by cluster (inserted file by file): every query's neighbourhood is ONE contiguous stretch
    of ~1,000 rows near the END of the table. The threshold stages walk the tiles in a spread-out order, so the result is
    the oracle's and no query overflows, let alone reaches the float64 scan (r1: contiguous stages)."""
    rng = np.random.default_rng(1000 + k)
    corpus = _clustered(rng, 60, 1000, 256)
    eng = eng_factory(256)
    eng.index_load(corpus)
    r = rng.integers(50_000, 60_000, 40)
    queries = (corpus[r] + 0.5 * rng.standard_normal((40, 256)) / np.sqrt(256)).astype(np.float32)
    st = check(eng, corpus, queries, k)
    # scores of foreign rows are correlated inside a cluster, so an unlucky 8-tile sample can leave the first threshold
    # loose for one query: such an overflow is repaired by the second MFMA pass, never by the float64 scan
    assert st["exact_scan"] == 0 and st["overflowed"] <= 2 and st["second_pass"] == st["overflowed"], st


def test_sorted_table_needs_no_fallback(eng_factory):
    """Topic-sorted table: a drift along one axis grows with the row number, so the rows most similar to the queries are
    all at the end of the table."""
    rng = np.random.default_rng(77)
    N, D = 70_000, 256
    u = rng.standard_normal(D)
    u /= np.linalg.norm(u)
    corpus = (rng.standard_normal((N, D)) / np.sqrt(D) + (3.0 * np.arange(N) / N - 1.5)[:, None] * u[None, :]).astype(np.float32)
    eng = eng_factory(D)
    eng.index_load(corpus)
    queries = (corpus[rng.integers(N - 5000, N, 50)] + 0.3 * rng.standard_normal((50, D)) / np.sqrt(D)).astype(np.float32)
    st = check(eng, corpus, queries, 100)
    assert st["exact_scan"] == 0 and st["second_pass"] == st["overflowed"] <= 2, st


def test_contiguous_tenants_walk_only_their_tiles(eng_factory):
    """`WHERE agent_id = %s` (rag/document_store.py:457) on a table exported tenant by tenant: the tenant's rows are one
    contiguous range; the search walks only the tiles that hold them. Also: a tenant smaller than k, a tenant id that owns
    no row, tenants whose ranges do not start on a tile boundary."""
    rng = np.random.default_rng(78)
    N, D, k = 50_000, 256, 20
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    bounds = [0, 7, 1000, 1300, 20_011, 20_500, 41_000, 49_990, N]                 # tenant t owns rows [bounds[t], bounds[t+1])
    tenants = np.zeros(N, dtype=np.int32)
    for t in range(len(bounds) - 1):
        tenants[bounds[t]:bounds[t + 1]] = t
    eng = eng_factory(D)
    eng.index_load(corpus)
    eng.set_tenants(tenants)
    for t in (0, 3, 5, 7, 6):
        rows = np.arange(bounds[t], bounds[t + 1])
        queries = (corpus[rng.choice(rows, 9)] + 0.5 * rng.standard_normal((9, D))).astype(np.float32)
        st = check(eng, corpus, queries, k, tenants, t)
        assert st["exact_scan"] == 0 and st["second_pass"] == st["overflowed"] <= 1, (t, st)
    ids, rows, sc = eng.dense_topk(corpus[:3], k, tenant=99)                       # unknown tenant: empty result
    assert (ids == -1).all() and (rows == -1).all() and (sc == 0).all()
    eng.set_tenants(None)


def test_second_pass_replaces_the_exact_scan(eng_factory, monkeypatch):
    """Forced overflow: one threshold stage over 300k rows with a threshold drawn from 2048 (option stage_growth) emits ~15k
    keys per query into the 4096-entry buffer. The select that follows still tightens tau from what was kept, and the second
    MFMA pass at that tau recovers every query - no float64 scan - with the oracle's result. With the second pass switched
    off the same queries take the scan path (in rounds of 256 flagged queries) and give the same answer."""
    rng = np.random.default_rng(79)
    N, D, Q, k = 300_000, 64, 200, 100          # the second pass takes up to 256 overflowed queries per search
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    queries = planted_queries(rng, corpus, Q)
    eng = eng_factory(D)
    eng.index_load(corpus)
    eng.set_option("stage_growth", 100000)
    st = check(eng, corpus, queries, k)
    assert st["overflowed"] >= Q and st["second_pass"] == Q and st["exact_scan"] == 0, st
    eng.set_option("no_second_pass", 1)
    st = check(eng, corpus, queries[:40], k)
    assert st["exact_scan"] == 40 and st["second_pass"] == 0, st


def test_exact_scan_rounds_beyond_256_flagged_queries(eng_factory, monkeypatch):
    """The float64 scan handles flagged queries in rounds of 256 with bounded scratch: 300 queries forced through it."""
    rng = np.random.default_rng(80)
    corpus = rng.standard_normal((9000, 128)).astype(np.float32)
    queries = planted_queries(rng, corpus, 300)
    eng = eng_factory(128)
    eng.index_load(corpus)
    eng.set_option("force_level", 2)
    st = check(eng, corpus, queries, 20)
    assert st["exact_scan"] == 300, st


def test_empty_index_searches_to_empty_results(eng_factory):
    """An index with zero rows is a valid state (a tenant's table before its first upload): every slot comes back -1 / 0.0,
    as `ORDER BY ... LIMIT k` over an empty table returns no rows (the zero-row branch could never run)."""
    eng = eng_factory(128)
    eng.index_load(np.zeros((0, 128), dtype=np.float32))
    ids, rows, sc = eng.dense_topk(np.ones((3, 128), dtype=np.float32), 5)
    assert (ids == -1).all() and (rows == -1).all() and (sc == 0).all()


def test_rccl_allgather_behind_the_c_abi_world_1():
    """rag_comm_*: librccl opened with dlopen, a one-rank communicator on this GPU, one all-gather on the caller's stream
    (recv[1][...] == send), destroy. More than one rank needs more than one GPU: the row-sharded composition itself is
    covered by the world-size-2 gloo tests and by the three-shards-on-one-GPU hybrid test."""
    import torch
    from optimized_rag_amd import RagEngine
    eng = RagEngine(dim=64, device=0)
    try:
        uid = eng.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.comm_init(0, 1, uid)
        send = torch.arange(2 * 7 * 5, dtype=torch.int64, device="cuda").reshape(2, 7, 5) * 3 - 11
        recv = torch.full((1, 2, 7, 5), -1, dtype=torch.int64, device="cuda")
        eng.comm_allgather_dev(send, recv)
        torch.cuda.synchronize()
        assert torch.equal(recv[0], send)
        eng.comm_destroy()
    finally:
        eng.close()


def test_tenant_with_nine_tiles_keeps_its_last_tile(eng_factory):
    """A tenant whose rows span 9 tiles is searched as ONE dense stage of 9 tiles (regression: the select after that stage
    was told 2048 slots and dropped the tenant's rows in its last tile; found by tests/test_property_gpu.py)."""
    rng = np.random.default_rng(2300)
    N, D = 6000, 64
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    tenants = np.full(N, 1, dtype=np.int32)
    tenants[700:3000] = 0                                          # rows 700..2999: tiles 2..11 -> 10 tiles, 2300 rows
    tenants[700:768] = 1                                           # -> rows 768..2999: tiles 3..11 = exactly 9 tiles
    eng = eng_factory(D)
    eng.index_load(corpus)
    eng.set_tenants(tenants)
    queries = planted_queries(rng, corpus[768:3000], 12, noise=0.2)
    st = check(eng, corpus, queries, 5, tenant_of_row=tenants, tenant=0)
    assert st["exact_scan"] == 0


@pytest.mark.parametrize("n_tiles", list(range(1, 21)) + [63, 64, 65, 71, 72, 73, 80])
def test_contiguous_tenant_spanning_n_tiles(eng_factory, n_tiles):
    """Stage boundaries of the tile schedule (8 dense tiles, then x8 growth, no tiny trailing stage) swept through a contiguous
    tenant that owns exactly n_tiles 256-row tiles (first and last tile only partly), against the float64 scan."""
    rng = np.random.default_rng(n_tiles)
    D = 64
    first = 256 * 3 + 17                                             # the tenant starts inside tile 3 ...
    last = 256 * (3 + n_tiles) - 40                                  # ... and ends inside tile 3 + n_tiles - 1
    N = last + 900
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    tenants = np.ones(N, dtype=np.int32)
    tenants[first:last] = 0
    eng = eng_factory(D)
    eng.index_load(corpus)
    eng.set_tenants(tenants)
    queries = planted_queries(rng, corpus[first:last], 9, noise=0.2)
    st = check(eng, corpus, queries, 7, tenant_of_row=tenants, tenant=0)
    assert st["exact_scan"] == 0
    eng.set_tenants(None)


@pytest.mark.parametrize("N", [1, 255, 256, 257, 2047, 2048, 2049, 2303, 2304, 2305, 2559, 2560, 2561, 4095, 4096, 4097, 16383, 16384, 16385,
                               18431, 18432, 18433, 20479, 20480, 20481, 131071, 131072, 131073, 147455, 147456, 147457])
def test_corpus_sizes_at_tile_and_stage_boundaries(eng_factory, N):
    rng = np.random.default_rng(N)
    D = 64
    corpus = rng.standard_normal((N, D)).astype(np.float32)
    queries = planted_queries(rng, corpus, 6, noise=0.2)
    queries[0] = corpus[N - 1]                                       # the very last row must be findable
    eng = eng_factory(D)
    eng.index_load(corpus)
    st = check(eng, corpus, queries, min(5, N))
    assert st["exact_scan"] == 0
