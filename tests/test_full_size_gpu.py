"""BASELINE.json configs[1] at FULL size (1M x 1536-d, 1024 queries, top-20) through size-independent properties:
planted neighbours, shard-merge invariance, proof counters, ordering, and a float64 spot check of a query subset."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, D, Q, K = 1_000_000, 1536, 1024, 20


@pytest.fixture(scope="module")
def world():
    import torch
    from optimized_rag_amd import RagEngine
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    corpus = torch.randn((N, D), generator=g, device=dev)
    corpus *= (0.5 + torch.rand((N, 1), generator=g, device=dev))            # un-normalised rows
    rows = torch.randint(0, N, (Q,), generator=torch.Generator().manual_seed(3))
    q = corpus[rows.to(dev)] * 0.37 + torch.randn((Q, D), generator=g, device=dev) * 0.3
    q[5] = corpus[rows[5]]                                                     # exact copy of a row: cosine 1.0
    whole = RagEngine(dim=D, device=0)
    whole.index_load(corpus)
    yield dict(torch=torch, dev=dev, corpus=corpus, q=q.contiguous(), rows=rows, whole=whole, RagEngine=RagEngine)
    whole.close()


def search(torch, eng, q, k):
    ids = torch.empty((q.shape[0], k), dtype=torch.int64, device=q.device)
    rows = torch.empty((q.shape[0], k), dtype=torch.int32, device=q.device)
    sc = torch.empty((q.shape[0], k), dtype=torch.float64, device=q.device)
    eng.dense_topk_dev(q, k, ids, rows, sc)
    torch.cuda.synchronize()
    return ids, sc


def test_planted_neighbours_order_and_proof(world):
    torch = world["torch"]
    ids, sc = search(torch, world["whole"], world["q"], K)
    ids_h, sc_h = ids.cpu().numpy(), sc.cpu().numpy()
    assert (ids_h[:, 0] == world["rows"].numpy()).all()                      # the planted row is the nearest neighbour
    assert abs(sc_h[5, 0] - 1.0) < 1e-12                                      # identical vector: cosine exactly ~1
    assert (np.diff(sc_h, axis=1) <= 0).all()                                 # sorted by cosine, descending
    assert (ids_h >= 0).all() and all(len(set(r)) == K for r in ids_h)        # k distinct rows
    st = world["whole"].dense_stats()
    assert st["proven_fast"] + st["proven_wide"] + st["exact_scan"] == Q and st["overflowed"] == 0
    world["ids"], world["sc"] = ids, sc


def test_shard_merge_invariance(world):
    """search(whole corpus) == merge(search(first 437,000 rows), search(the rest)): bit-identical ids and scores."""
    torch = world["torch"]
    cut = 437_000
    parts = []
    for lo, hi in ((0, cut), (cut, N)):
        e = world["RagEngine"](dim=D, device=0)
        e.index_load(world["corpus"][lo:hi].contiguous(), id_base=lo)
        parts.append(search(torch, e, world["q"], K))
        e.close()
    ids = torch.stack([p[0] for p in parts]).contiguous()
    sc = torch.stack([p[1] for p in parts]).contiguous()
    out_i = torch.empty((Q, K), dtype=torch.int64, device=world["dev"])
    out_s = torch.empty((Q, K), dtype=torch.float64, device=world["dev"])
    world["whole"].merge_topk_dev(ids, sc, out_i, out_s)
    torch.cuda.synchronize()
    if "ids" not in world:
        world["ids"], world["sc"] = search(torch, world["whole"], world["q"], K)
    assert torch.equal(out_i, world["ids"])
    assert torch.equal(out_s, world["sc"])                                     # same float64 kernel, same rows: identical bits


def test_float64_spot_check_of_a_query_subset(world):
    """8 queries: float32 BLAS shortlist of 200 on the host, re-scored in float64 = the oracle's exact top-20."""
    torch = world["torch"]
    from oracle import rag_oracle as O
    sel = [0, 5, 17, 300, 511, 512, 800, 1023]
    hc = world["corpus"].cpu().numpy()
    hq = world["q"][sel].cpu().numpy()
    unit = hc / np.linalg.norm(hc, axis=1, keepdims=True)
    s32 = (hq / np.linalg.norm(hq, axis=1, keepdims=True)) @ unit.T
    if "ids" not in world:
        world["ids"], world["sc"] = search(torch, world["whole"], world["q"], K)
    got_i, got_s = world["ids"].cpu().numpy(), world["sc"].cpu().numpy()
    for j, qi in enumerate(sel):
        short = np.argpartition(-s32[j], 200)[:200]
        exact = O.cosine_matrix(hq[j:j + 1], hc[short])[0]
        order = np.lexsort((short, -exact))[:K]
        np.testing.assert_array_equal(got_i[qi], short[order])
        np.testing.assert_allclose(got_s[qi], exact[order], atol=1e-9)


def _postings(world):
    """1M-doc synthetic postings + 1024 term queries, built once per module and loaded into world['whole']."""
    if "post" not in world:
        import bench_modes as BM
        from optimized_rag_amd.bm25 import Bm25Postings
        indptr, d, tf, dl, tok, doc_ptr = BM.synthetic_csr(N, 100_000, 120)
        post = Bm25Postings(indptr, d, tf, dl, Bm25Postings.idf_table(np.diff(indptr).clip(min=0), N), float(dl.sum()) / N)
        post.idf[np.diff(indptr) == 0] = 0.0
        post.load(world["whole"])
        rng = np.random.default_rng(7)
        ptr, terms = [0], []
        for i in range(Q):
            di = int(rng.integers(0, N))
            toks = tok[doc_ptr[di]:doc_ptr[di + 1]]
            n = int(rng.integers(4, 13))
            terms.extend(int(x) for x in (rng.choice(toks, n) if len(toks) else [0] * n))
            ptr.append(len(terms))
        world["post"] = (post, np.asarray(ptr, np.int32), np.asarray(terms, np.int32))
    return world["post"]


def test_bm25_full_size_staged_equals_exhaustive_select(world, monkeypatch):
    """BASELINE.json configs[2] at FULL size (1M docs, ~9.5e7 postings, 489 doc ranges, 1024 queries, top-100): the staged-
    threshold BM25 path must return bit-identical ids and scores to the path that runs the exact per-range select on every range,
    six sampled queries must equal the CSR oracle (BM25Okapi.get_scores / max, stable top-100) bit for bit, the scores are
    max-normalised and sorted, and the sharded pipeline property holds: scoring two doc partitions with the global statistics
    and merging == scoring the whole corpus."""
    import torch
    eng = world["whole"]
    post, ptr, terms = _postings(world)
    ptr_d, terms_d = torch.from_numpy(ptr).cuda(), torch.from_numpy(terms).cuda()
    pool = 100

    def run(e):
        ids = torch.empty((Q, pool), dtype=torch.int64, device="cuda")
        sc = torch.empty((Q, pool), dtype=torch.float64, device="cuda")
        mx = torch.empty((Q,), dtype=torch.float64, device="cuda")
        e.bm25_topk_dev(ptr_d, terms_d, pool, ids, None, sc, mx)
        torch.cuda.synchronize()
        return ids.cpu().numpy(), sc.cpu().numpy(), mx.cpu().numpy()

    ids_s, sc_s, mx_s = run(eng)                                   # staged threshold (default)
    eng.set_option("bm25_no_staging", 1)
    try:
        ids_e, sc_e, mx_e = run(eng)                               # exact select on all 489 ranges
    finally:
        eng.set_option("bm25_no_staging", 0)
    np.testing.assert_array_equal(ids_s, ids_e)
    np.testing.assert_array_equal(sc_s, sc_e)
    np.testing.assert_array_equal(mx_s, mx_e)
    assert (sc_s[:, 0] == 1.0).all() and (np.diff(sc_s, axis=1) <= 0).all() and (ids_s >= 0).all()
    from oracle import rag_oracle as O
    for qi in (0, 1, 255, 256, 700, 1023):
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, terms[ptr[qi]:ptr[qi + 1]].tolist())
        top = O.stable_topk_desc(raw, pool)
        np.testing.assert_array_equal(ids_s[qi], top)
        np.testing.assert_array_equal(sc_s[qi], raw[top] / (raw.max() if raw.max() > 0 else 1.0))
        assert mx_s[qi] == (raw.max() if raw.max() > 0 else 1.0)
    # two doc partitions with global statistics, raw scores, merged on the device == the whole corpus
    cut = 437_000
    parts = []
    for lo, hi in ((0, cut), (cut, N)):
        e = world["RagEngine"](dim=D, device=0)
        e.index_reserve(hi - lo, id_base=lo)
        e.index_append(world["corpus"][lo:hi].contiguous())          # row-aligned ids for the BM25 doc numbers
        post.shard(lo, hi).load(e)
        e.bm25_set_normalize(False)
        parts.append(run(e))
        e.close()
    ids = torch.from_numpy(np.stack([p[0] for p in parts])).cuda()
    sc = torch.from_numpy(np.stack([p[1] for p in parts])).cuda()
    oi = torch.empty((Q, pool), dtype=torch.int64, device="cuda")
    os_ = torch.empty((Q, pool), dtype=torch.float64, device="cuda")
    eng.merge_topk_dev(ids, sc, oi, os_)
    torch.cuda.synchronize()
    merged = os_.cpu().numpy()
    np.testing.assert_array_equal(oi.cpu().numpy(), ids_s)
    np.testing.assert_array_equal(merged / np.where(merged[:, :1] > 0, merged[:, :1], 1.0), sc_s)


def test_retrieve_rerank_full_size_vs_oracle_on_sampled_queries(world):
    """BASELINE.json configs[3] at FULL size through the one-call entry: 1M rows + ~9.5e7 postings + a 224-token passage
    store, 256 queries, hybrid top-100 -> MiniLM-L-6 cross-encoder (L = 256, 6 layers, 25,600 pairs = 3 activation
    chunks) -> top-20. Every query: slots filled, scores sorted, ids drawn from its candidate list. Two sampled queries
    get the whole oracle composition: float64 dense top-100, CSR BM25 top-100, RRF ranks (candidate list bit-exact),
    'longest_first' pair assembly, float64 BERT forward of all 100 pairs, sigmoid, stable sort."""
    import torch
    from oracle import bert_oracle as B
    from oracle import rag_oracle as O
    from optimized_rag_amd.cross_encoder import flatten_state_dict
    eng = world["whole"]
    post, ptr, terms = _postings(world)
    Qr, pool, k, L, Ld, Lq = 256, 100, 20, 256, 224, 16
    cfg = B.minilm_config()
    w = B.seeded_weights(cfg, 2024)
    eng.ce_load(cfg, flatten_state_dict(w, cfg["layers"]))
    tok = torch.randint(1000, cfg["vocab_size"], (N, Ld), generator=torch.Generator().manual_seed(5), dtype=torch.int32).numpy()
    tok_len = torch.randint(96, Ld + 1, (N,), generator=torch.Generator().manual_seed(6), dtype=torch.int32).numpy()
    eng.tokens_load(tok, tok_len)
    q_tok = torch.randint(1000, cfg["vocab_size"], (Qr, Lq), generator=torch.Generator().manual_seed(8), dtype=torch.int32)
    q_len = torch.full((Qr,), Lq, dtype=torch.int32)
    q_emb = world["q"][:Qr].contiguous()
    ids, sc, lg, cand = eng.retrieve_rerank_dev(q_emb, q_tok.cuda(), q_len.cuda(), pool, k, term_ptr=torch.from_numpy(ptr[:Qr + 1]).cuda(),
                                                terms=torch.from_numpy(terms).cuda(), L_pair=L)
    torch.cuda.synchronize()
    ids, sc, lg, cand = ids.cpu().numpy(), sc.cpu().numpy(), lg.cpu().numpy(), cand.cpu().numpy()
    assert (ids >= 0).all() and (cand >= 0).all() and (np.diff(sc, axis=1) <= 0).all()
    assert all(set(ids[q]) <= set(cand[q]) and len(set(ids[q])) == k for q in range(Qr))
    np.testing.assert_allclose(sc, 1.0 / (1.0 + np.exp(-lg.astype(np.float64))), atol=1e-15)
    # ---- oracle composition for two queries --------------------------------------------------------------------
    hc = world["corpus"].cpu().numpy()
    unit = hc / np.linalg.norm(hc, axis=1, keepdims=True)
    for qi in (3, 200):
        hq = q_emb[qi:qi + 1].cpu().numpy()
        s32 = ((hq / np.linalg.norm(hq)) @ unit.T)[0]
        short = np.argpartition(-s32, 600)[:600]
        exact = O.cosine_matrix(hq, hc[short])[0]
        d_rows = short[np.lexsort((short, -exact))[:pool]]
        qt = terms[ptr[qi]:ptr[qi + 1]].tolist()
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, qt)
        b_rows = O.stable_topk_desc(raw, pool)
        okeys, _, _ = O.rrf_fuse([[int(r) for r in d_rows], [int(r) for r in b_rows]], k=60, top_k=pool)
        assert cand[qi].tolist() == okeys                                            # candidate list: bit-exact
        pid = np.zeros((pool, L), dtype=np.int64)
        ptt = np.zeros((pool, L), dtype=np.int64)
        plen = np.zeros(pool, dtype=np.int64)
        for j, r in enumerate(okeys):
            ql, dl = O.longest_first_lengths(Lq, int(tok_len[r]), L - 3)
            row = [101] + q_tok[qi, :ql].tolist() + [102] + tok[r, :dl].tolist() + [102]
            pid[j, :len(row)] = row
            ptt[j, ql + 2:len(row)] = 1
            plen[j] = len(row)
        ologit = B.forward_logits(w, cfg, pid, ptt, plen, fast_erf=True)
        oscore = np.array([O.sigmoid(float(x)) for x in ologit])
        order = sorted(range(pool), key=lambda j: -oscore[j])
        np.testing.assert_allclose(sc[qi], oscore[order[:k]], atol=1e-3)
        np.testing.assert_allclose(lg[qi], ologit[order[:k]], atol=4e-3)
        gaps = np.abs(np.diff(oscore[order[:k + 1]]))
        if gaps.min() > 2e-3:
            assert ids[qi].tolist() == [okeys[j] for j in order[:k]]


@pytest.mark.parametrize("kind", ["clustered", "tenant-contiguous"])
def test_structured_row_order_full_size(world, kind):
    """VERDICT r1 item 4 at 1M x 1536: (a) 1,000 clusters stored cluster by cluster, 1024 queries planted in the last fifth
    of the table; (b) 100 tenants stored contiguously, the batch filtered on tenant 97. No query may overflow or reach the
    float64 scan, the planted row is rank 1, and a float64 spot check of 6 queries reproduces ids and scores."""
    import bench as BE
    from oracle import rag_oracle as O
    torch = world["torch"]
    dev = world["dev"]
    eng = world["RagEngine"](dim=D, device=0)
    try:
        eng.index_reserve(N)
        parts = []
        for c in range(N // BE.CHUNK_ROWS):
            blk = BE.gen_chunk(c, BE.CHUNK_ROWS, dev, kind, N)
            eng.index_append(blk)
            parts.append(blk.cpu())
        q, planted = BE.gen_queries(Q, N, N // BE.CHUNK_ROWS, BE.CHUNK_ROWS, dev, kind)
        tenant, t_lo, t_hi = -1, 0, N
        if kind == "tenant-contiguous":
            eng.set_tenants(BE.bench_tenants(N))
            tenant = BE.BENCH_TENANT
            t_lo, t_hi = tenant * N // BE.N_TENANTS, (tenant + 1) * N // BE.N_TENANTS
        ids = torch.empty((Q, K), dtype=torch.int64, device=dev)
        sc = torch.empty((Q, K), dtype=torch.float64, device=dev)
        eng.dense_topk_dev(q, K, ids, None, sc, tenant=tenant)
        torch.cuda.synchronize()
        st = eng.dense_stats()
        assert st["exact_scan"] == 0 and st["second_pass"] == st["overflowed"] <= 8 and st["proven_fast"] + st["proven_wide"] == Q, st
        ids_h, sc_h = ids.cpu().numpy(), sc.cpu().numpy()
        assert (ids_h[:, 0] == planted.numpy()).all() and (np.diff(sc_h, axis=1) <= 0).all()
        assert ((ids_h >= t_lo) & (ids_h < t_hi)).all()
        hc = torch.cat(parts).numpy()[t_lo:t_hi]
        for qi in (0, 1, 511, 512, 777, 1023):
            hq = q[qi:qi + 1].cpu().numpy()
            s32 = (hq @ hc.T)[0]
            short = np.argpartition(-s32, 400)[:400]
            exact = O.cosine_matrix(hq, hc[short])[0]
            order = np.lexsort((short, -exact))[:K]
            np.testing.assert_array_equal(ids_h[qi], short[order] + t_lo)
            np.testing.assert_allclose(sc_h[qi], exact[order], atol=1e-9)
    finally:
        eng.close()


def test_index_level_linear_hybrid_full_size(world):
    """rag_hybrid_linear_dev at 1M rows x 300 queries (two 256-query sub-batches): properties that do not need the CPU
    oracle - hybrid == (alpha*sem + beta*kw) + gamma*tmp recomputed from the returned components bit for bit, scores sorted,
    distinct rows, keyword in [0, 1] with the per-query maximum reachable - plus 3 queries checked against a float64
    recomputation over a shortlist that provably contains the top-k (every row whose upper bound alpha + beta*kw + gamma*t
    could reach the k-th score)."""
    import torch
    from oracle import rag_oracle as O
    eng = world["whole"]
    post, ptr, terms = _postings(world)
    Ql, k, a, b, g = 300, 20, 0.55, 0.35, 0.10
    rng = np.random.default_rng(11)
    temporal = np.where(rng.uniform(size=N) < 0.3, 0.15 * 0.5 ** (rng.uniform(0, 200, N) / 30.0), 0.0)
    eng.set_temporal(temporal)
    q = world["q"][:Ql].contiguous()
    out = eng.hybrid_linear_dev(q, torch.from_numpy(ptr[:Ql + 1]).cuda(), torch.from_numpy(terms).cuda(), k, a, b, g)
    torch.cuda.synchronize()
    got = {key: v.cpu().numpy() for key, v in out.items()}
    assert (got["rows"] >= 0).all() and all(len(set(r)) == k for r in got["rows"])
    assert (np.diff(got["hybrid"], axis=1) <= 0).all()
    assert ((a * got["semantic"] + b * got["keyword"]) + g * got["temporal"] == got["hybrid"]).all()
    assert (got["keyword"] >= 0).all() and (got["keyword"] <= 1).all()
    np.testing.assert_array_equal(got["temporal"], temporal[got["rows"]])
    hc = world["corpus"].cpu().numpy()
    unit = hc / np.linalg.norm(hc, axis=1, keepdims=True)
    for qi in (0, 255, 299):
        raw = O.bm25_scores_csr(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, terms[ptr[qi]:ptr[qi + 1]].tolist())
        kw = raw / (raw.max() if raw.max() > 0 else 1.0)
        hq = q[qi:qi + 1].cpu().numpy()
        s32 = ((hq / np.linalg.norm(hq)) @ unit.T)[0].astype(np.float64)
        approx = (a * s32 + b * kw) + g * temporal
        short = np.nonzero(approx >= np.sort(approx)[-k] - 1e-3)[0]                 # float32 cosine error << 1e-3
        exact = O.cosine_matrix(hq, hc[short])[0]
        hyb = (a * exact + b * kw[short]) + g * temporal[short]
        order = np.lexsort((short, -hyb))[:k]
        np.testing.assert_array_equal(got["rows"][qi], short[order])
        np.testing.assert_allclose(got["hybrid"][qi], hyb[order], atol=1e-12)
        assert got["keyword"][qi].tolist() == kw[short[order]].tolist()
    eng.set_temporal(None)
