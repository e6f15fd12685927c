"""TEST / MEASUREMENT INFRASTRUCTURE — CPU baseline legs of bench.py ("port" kind). Not product code.

The reference's hot path on the GPU box's host cores, restated (neither Postgres/pgvector, rank-bm25 nor
sentence-transformers exist there; BASELINE.md section 2 lists these legs):
  dense_topk_blas            `ORDER BY embedding <=> q LIMIT k` (/root/reference/rag/document_store.py:448-460) as a float32
                             BLAS exact scan + top-k on all cores (same semantics as oracle/rag_oracle.py::dense_topk)
  bm25_topk_numpy            BM25Okapi.get_scores + /max + top-k (/root/reference/rag/retrieval.py:324-347) as the numpy
                             restatement over CSR postings (oracle/rag_oracle.py::bm25_scores_csr)
  bert_cpu_pairs             CrossEncoder.predict (/root/reference/rag/reranker.py:355): torch-CPU
                             BertForSequenceClassification, batch 32, padded to the longest pair of the batch
  python_loop_*              the LITERAL pure-Python loops of the reference (rag/retrieval.py:253-256,362-371;
                             rag/consistency_checker.py:169-189; rag/nodes/helpers.py:229-252), one core
Every function is timed on a BOUNDED sample and says what the sample was.
"""
import math
import os
import time

import numpy as np


def effective_cores():
    """CPU threads this process may really use: min(affinity mask, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def dense_topk_blas(corpus_unit, queries_unit, k, threads=None):
    """corpus_unit [N,D] float32 unit rows (torch CPU tensor), queries_unit [Q,D]. Returns (idx [Q,k], seconds)."""
    import torch
    threads = threads or effective_cores()
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    s = queries_unit @ corpus_unit.T                        # [Q,N] float32, BLAS on all cores
    top = torch.topk(s, k, dim=1, largest=True, sorted=True)
    dt = time.perf_counter() - t0
    return top.indices.numpy(), top.values.numpy(), dt, threads


def bm25_topk_numpy(indptr, doc, tf, doc_len, idf, avgdl, term_ptr, terms, k, k1=1.5, b=0.75):
    """numpy BM25Okapi over CSR for a sample of queries: scores of ALL docs (float64), / max, stable top-k.
    Returns (rows [Q,k], seconds). One core (numpy fancy indexing), as rank-bm25 itself runs."""
    from oracle import rag_oracle as O
    Q = len(term_ptr) - 1
    rows = np.empty((Q, k), dtype=np.int64)
    t0 = time.perf_counter()
    for q in range(Q):
        raw = O.bm25_scores_csr(indptr, doc, tf, doc_len, idf, avgdl, terms[term_ptr[q]:term_ptr[q + 1]].tolist(), k1, b)
        mx = raw.max() if raw.max() > 0 else 1.0
        raw /= mx
        part = np.argpartition(-raw, k)[:4 * k] if raw.shape[0] > 4 * k else np.arange(raw.shape[0])
        # stable order (score desc, doc asc) inside a shortlist that certainly holds the top-k unless a tie plateau is
        # wider than the shortlist; the GPU-side parity tests cover plateaus, this leg is only timed
        order = np.lexsort((part, -raw[part]))[:k]
        rows[q] = part[order]
    return rows, time.perf_counter() - t0


def hf_state_dict(cfg, tensors):
    """rag_ce_load_host's flat tensor order -> HF BertForSequenceClassification state-dict names."""
    from optimized_rag_amd.cross_encoder import LAYER_KEYS
    names = ["bert.embeddings.word_embeddings.weight", "bert.embeddings.position_embeddings.weight",
             "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight", "bert.embeddings.LayerNorm.bias"]
    for l in range(cfg["layers"]):
        names += [f"bert.encoder.layer.{l}.{k}" for k in LAYER_KEYS]
    names += ["bert.pooler.dense.weight", "bert.pooler.dense.bias", "classifier.weight", "classifier.bias"]
    assert len(names) == len(tensors)
    return dict(zip(names, tensors))


def bert_cpu_pairs(cfg, tensors, ids, tt, lens, batch=32, threads=None):
    """torch-CPU BertForSequenceClassification (what sentence-transformers runs without a GPU), batches of 32 pairs padded
    to the longest pair of the batch. Returns (logits [P], seconds, threads)."""
    import torch
    import transformers as tr
    threads = threads or effective_cores()
    torch.set_num_threads(threads)
    hf = tr.BertForSequenceClassification(tr.BertConfig(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"],
        num_attention_heads=cfg["heads"], intermediate_size=cfg["ffn"], max_position_embeddings=cfg["max_pos"],
        type_vocab_size=cfg.get("type_vocab", 2), hidden_act="gelu", layer_norm_eps=cfg.get("eps", 1e-12), num_labels=1)).eval()
    sd = hf.state_dict()
    for name, t in hf_state_dict(cfg, tensors).items():
        sd[name].copy_(torch.from_numpy(np.asarray(t)))
    P = ids.shape[0]
    out = np.empty((P,), dtype=np.float32)
    ids_t, tt_t = torch.from_numpy(np.asarray(ids, dtype=np.int64)), torch.from_numpy(np.asarray(tt, dtype=np.int64))
    with torch.no_grad():
        hf(input_ids=ids_t[:2, :16], token_type_ids=tt_t[:2, :16])                  # one-time lazy initialisation, untimed
        t0 = time.perf_counter()
        for b0 in range(0, P, batch):
            sl = slice(b0, min(P, b0 + batch))
            L = int(np.max(lens[sl]))
            mask = torch.from_numpy((np.arange(L)[None, :] < np.asarray(lens[sl])[:, None]).astype(np.int64))
            out[sl] = hf(input_ids=ids_t[sl, :L], token_type_ids=tt_t[sl, :L], attention_mask=mask).logits[:, 0].numpy()
        dt = time.perf_counter() - t0
    return out, dt, threads


def _cos_loop(v1, v2):
    """rag/retrieval.py:362-371, verbatim semantics: three generator passes per pair."""
    dot = sum(a * b for a, b in zip(v1, v2))
    m1 = math.sqrt(sum(a * a for a in v1))
    m2 = math.sqrt(sum(b * b for b in v2))
    if m1 == 0 or m2 == 0:
        return 0.0
    return dot / (m1 * m2)


def python_loop_cosine_rate(dim=1536, n=2000):
    """pairs/s of the literal reference cosine (kept for tools/ and older profiles)."""
    rng = np.random.default_rng(0)
    q = [float(x) for x in rng.standard_normal(dim)]
    docs = [[float(x) for x in rng.standard_normal(dim)] for _ in range(n)]
    t0 = time.perf_counter()
    for d in docs:
        _cos_loop(q, d)
    return n / (time.perf_counter() - t0)


def python_loop_semantic_scan(n=4096, dim=1536):
    """hybrid_search's semantic leg (rag/retrieval.py:253-256): one query against n documents, pure Python. seconds."""
    rng = np.random.default_rng(0)
    q = [float(x) for x in rng.standard_normal(dim)]
    docs = [[float(x) for x in row] for row in rng.standard_normal((n, dim))]
    t0 = time.perf_counter()
    scores = [_cos_loop(q, d) for d in docs]
    dt = time.perf_counter() - t0
    return dt, scores


def python_loop_consistency(embs, doc_idx, threshold=0.85):
    """ConsistencyChecker._find_contradictions' pair loop (rag/consistency_checker.py:169-189) on given claim embeddings:
    every i < j from different documents costs one literal cosine. seconds."""
    vecs = [[float(x) for x in e] for e in embs]
    t0 = time.perf_counter()
    hits = 0
    for i in range(len(vecs)):
        for j in range(i + 1, len(vecs)):
            if doc_idx[i] == doc_idx[j]:
                continue
            if _cos_loop(vecs[i], vecs[j]) >= threshold:
                hits += 1
    return time.perf_counter() - t0, hits


def python_loop_mmr(q, embs, k, lam):
    """apply_mmr's greedy loop (rag/nodes/helpers.py:229-252) with literal cosines. (seconds, picks)"""
    qv = [float(x) for x in q]
    vecs = [[float(x) for x in e] for e in embs]
    t0 = time.perf_counter()
    selected, remaining = [], list(range(len(vecs)))
    while len(selected) < k and remaining:
        best, best_s = None, -float("inf")
        for i in remaining:
            rel = _cos_loop(qv, vecs[i])
            ms = max((_cos_loop(vecs[i], vecs[s]) for s in selected), default=0.0)
            s = lam * rel - (1 - lam) * ms
            if s > best_s:
                best, best_s = i, s
        selected.append(best)
        remaining.remove(best)
    return time.perf_counter() - t0, selected


def pgvector_probe_and_time(corpus, queries, k, sample_rows=200_000, timeout_s=120):
    """SURVEY 8d baseline (2): when - and only when - a `postgres` server binary WITH the `vector` extension is found on this
    box at run time, time the reference's real statement, `ORDER BY embedding <=> $1 LIMIT k` as a sequential scan
    (/root/reference/rag/document_store.py:448-460 without the HNSW index), in a throw-away cluster under /tmp. Absent in the
    build container and on the GPU boxes of this pool: the function then says what it looked for. corpus [N, D] / queries
    [Q, D] float32 numpy arrays; a bounded prefix of the corpus is loaded. Returns a dict for the bench line."""
    import shutil
    import subprocess
    import tempfile
    need = {b: shutil.which(b) for b in ("postgres", "initdb", "pg_ctl", "psql", "pg_config")}
    missing = [b for b, p in need.items() if p is None]
    if missing:
        return {"found": False, "looked_for": sorted(need), "missing": missing}
    try:
        share = subprocess.run([need["pg_config"], "--sharedir"], capture_output=True, text=True, timeout=10).stdout.strip()
    except (OSError, subprocess.SubprocessError) as e:
        return {"found": False, "error": f"pg_config: {e}"}
    if not os.path.exists(os.path.join(share, "extension", "vector.control")):
        return {"found": False, "postgres": need["postgres"], "missing": ["extension/vector.control (pgvector)"]}
    n = min(int(sample_rows), corpus.shape[0])
    d = tempfile.mkdtemp(prefix="rag_pg_")
    sock, data = d, os.path.join(d, "data")
    run = lambda cmd, **kw: subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, **kw)
    psql = [need["psql"], "-h", sock, "-d", "postgres", "-v", "ON_ERROR_STOP=1", "-qAt"]
    try:
        r = run([need["initdb"], "-D", data, "-A", "trust"])
        if r.returncode:
            return {"found": True, "error": "initdb: " + r.stderr[-300:]}
        r = run([need["pg_ctl"], "-D", data, "-o", f"-k {sock} -c listen_addresses='' -c shared_buffers=1GB -c max_parallel_workers_per_gather={effective_cores()}",
                 "-w", "start"])
        if r.returncode:
            return {"found": True, "error": "pg_ctl start: " + r.stderr[-300:]}
        D = corpus.shape[1]
        r = run(psql + ["-c", f"CREATE EXTENSION vector; CREATE TABLE c (id bigint, embedding vector({D}));"])
        if r.returncode:
            return {"found": True, "error": "create: " + r.stderr[-300:]}
        # COPY in chunks of 2,000 rows (~40 MB of text each): the whole sample as one Python string would be several GB
        for b0 in range(0, n, 2000):
            rows = "\n".join("%d\t[%s]" % (i, ",".join("%.7g" % x for x in corpus[i])) for i in range(b0, min(n, b0 + 2000)))
            r = subprocess.run(psql + ["-c", "COPY c FROM STDIN"], input=rows, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode:
                return {"found": True, "error": "copy: " + r.stderr[-300:]}
        nq = min(8, queries.shape[0])
        t0 = time.perf_counter()
        for qi in range(nq):
            lit = "[" + ",".join("%.7g" % x for x in queries[qi]) + "]"
            r = run(psql + ["-c", f"SELECT id FROM c ORDER BY embedding <=> '{lit}'::vector LIMIT {int(k)};"])
            if r.returncode:
                return {"found": True, "error": "query: " + r.stderr[-300:]}
        dt = time.perf_counter() - t0
        return {"found": True, "value": round(nq / dt, 3), "unit": "queries/sec", "kind": "reference",
                "sample": f"{nq} queries, sequential `ORDER BY embedding <=> q LIMIT {k}` over the first {n} rows ({dt:.2f}s, psql round trips included)"}
    except (OSError, subprocess.SubprocessError) as e:
        return {"found": True, "error": f"{type(e).__name__}: {e}"}
    finally:
        try:                                            # a stuck or missing server must never cost the bench line
            subprocess.run([need["pg_ctl"], "-D", data, "-m", "immediate", "stop"], capture_output=True, timeout=60)
        except (OSError, subprocess.SubprocessError):
            pass
        shutil.rmtree(d, ignore_errors=True)
