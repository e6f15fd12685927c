"""On-disk corpus shard: the bulk-load format between the reference's Postgres tables and the HBM-resident index
(SURVEY.md section 8f.2 — "the data formats either side of the path").

Source tables (read-only facts about the reference, nothing is executed against Postgres here):
  document_chunks(id, document_id, agent_id TEXT, chunk_index, content TEXT, embedding vector(D), metadata JSONB,
                  created_at TIMESTAMP)                       /root/reference/rag/document_store.py:210-221
  archival_memory(id, agent_id, content, embedding vector(D), metadata, created_at)
                                                              /root/reference/database/operations.py:40-44
A row's `embedding` arrives either as a float sequence (pgvector's psycopg2 adapter) or as pgvector's TEXT form
`'[0.1,0.2,...]'` (what `SELECT embedding::text` / a plain cursor returns) — both are accepted.

Shard directory (everything little-endian, memory-mappable, rows in export order):
  meta.json          {"version", "n_rows", "dim", "tenants": {agent_id: int}, "has_bm25", "token_len"}
  embeddings.npy     [N, D] float32        the `vector(D)` column, bit-for-bit (pgvector stores float4)
  ids.npy            [N]    int64          the table's primary key (doc ids the search returns)
  tenants.npy        [N]    int32          agent_id -> tenant number (the `WHERE agent_id = %s` filter, on device)
  created_at.npy     [N]    float64        POSIX seconds, NaN = NULL (temporal boost, retrieval.py:266-292)
  rows.jsonl + rows.idx.npy                {content, metadata, filename?, file_type?} per row + byte offsets
  bm25.npz           indptr/doc/tf/doc_len/idf/avgdl/vocab   (optional) term-major CSR built by Bm25Postings.from_corpus
  tokens.npy, token_lens.npy               (optional) [N, L] int32 passage token ids for the cross-encoder

Loading streams `chunk_rows` rows at a time from the memory map into rag_index_reserve/rag_index_append_host, so a
12.5M-row (77 GB) shard never sits in host memory twice; for one-process-per-GPU runs every rank loads its
contiguous row range `[begin, end)` of the same directory, with the BM25 postings sliced by Bm25Postings.shard.
"""
import json
import os
from datetime import datetime, timezone
from typing import Any, Dict, Iterable, Optional

import numpy as np

from .bm25 import Bm25Postings

FORMAT_VERSION = 1


def parse_pgvector_text(text: str, dim: Optional[int] = None) -> np.ndarray:
    """pgvector's text output `[x1,x2,...]` -> float32 vector (float4 values print with enough digits to round-trip)."""
    s = text.strip()
    if not (s.startswith("[") and s.endswith("]")):
        raise ValueError("not a pgvector text literal: %r" % (text[:40],))
    body = s[1:-1].strip()
    v = np.array([float(x) for x in body.split(",")], dtype=np.float32) if body else np.zeros(0, dtype=np.float32)
    if dim is not None and v.shape[0] != dim:
        raise ValueError("expected %d dimensions, got %d" % (dim, v.shape[0]))
    return v


def format_pgvector_text(vec) -> str:
    """float32 vector -> a literal pgvector accepts for `%s::vector`; every value round-trips to the same float4."""
    return "[" + ",".join(repr(float(x)) for x in np.asarray(vec, dtype=np.float32)) + "]"


def _epoch(ts) -> float:
    """created_at (datetime, ISO string, number or None) -> POSIX seconds; naive datetimes are taken as UTC."""
    if ts is None:
        return float("nan")
    if isinstance(ts, (int, float)):
        return float(ts)
    if isinstance(ts, str):
        try:
            ts = datetime.fromisoformat(ts.replace("Z", "+00:00"))
        except ValueError:
            return float("nan")
    if ts.tzinfo is None:
        ts = ts.replace(tzinfo=timezone.utc)
    return ts.timestamp()


class ShardWriter:
    """Streaming writer: `add()` one exported table row at a time, `close()` finalises the arrays and the BM25 CSR."""

    def __init__(self, path: str, dim: int = 1536):
        self.path, self.dim = path, dim
        os.makedirs(path, exist_ok=True)
        self._emb = open(os.path.join(path, "embeddings.raw"), "wb")
        self._rows = open(os.path.join(path, "rows.jsonl"), "wb")
        self._ids, self._tenants, self._created, self._offsets = [], [], [], [0]
        self._tenant_id: Dict[str, int] = {}
        self._contents = []

    def add(self, id: int, agent_id: str, content: str, embedding, metadata: Any = None, created_at=None,
            filename: Optional[str] = None, file_type: Optional[str] = None) -> None:
        v = parse_pgvector_text(embedding, self.dim) if isinstance(embedding, str) else np.asarray(embedding, dtype=np.float32)
        if v.shape != (self.dim,):
            raise ValueError("row %r: embedding has shape %r, expected (%d,)" % (id, v.shape, self.dim))
        self._emb.write(np.ascontiguousarray(v).tobytes())
        if isinstance(metadata, str):                                    # JSONB read through a plain cursor
            try:
                metadata = json.loads(metadata)
            except ValueError:
                metadata = {"raw": metadata}
        rec = {"content": content, "metadata": metadata or {}}
        if filename is not None:
            rec["filename"] = filename
        if file_type is not None:
            rec["file_type"] = file_type
        line = (json.dumps(rec, ensure_ascii=False, default=str) + "\n").encode("utf-8")
        self._rows.write(line)
        self._offsets.append(self._offsets[-1] + len(line))
        self._ids.append(int(id))
        self._tenants.append(self._tenant_id.setdefault(str(agent_id), len(self._tenant_id)))
        self._created.append(_epoch(created_at))
        self._contents.append(content)

    def close(self, build_bm25: bool = True, tokens=None, token_lens=None) -> str:
        self._emb.close()
        self._rows.close()
        n = len(self._ids)
        raw = os.path.join(self.path, "embeddings.raw")
        # wrap the raw float32 stream in a .npy header without a second copy in memory
        mm = np.lib.format.open_memmap(os.path.join(self.path, "embeddings.npy"), mode="w+", dtype=np.float32,
                                       shape=(n, self.dim))
        src = np.memmap(raw, dtype=np.float32, mode="r", shape=(n, self.dim)) if n else np.zeros((0, self.dim), np.float32)
        step = 65536
        for b in range(0, n, step):
            mm[b:b + step] = src[b:b + step]
        mm.flush()
        del mm, src
        os.remove(raw)
        np.save(os.path.join(self.path, "ids.npy"), np.asarray(self._ids, dtype=np.int64))
        np.save(os.path.join(self.path, "tenants.npy"), np.asarray(self._tenants, dtype=np.int32))
        np.save(os.path.join(self.path, "created_at.npy"), np.asarray(self._created, dtype=np.float64))
        np.save(os.path.join(self.path, "rows.idx.npy"), np.asarray(self._offsets, dtype=np.int64))
        if build_bm25 and n:
            p = Bm25Postings.from_corpus(self._contents)
            vocab = np.array(sorted(p.vocab, key=p.vocab.get), dtype=object)
            np.savez(os.path.join(self.path, "bm25.npz"), indptr=p.indptr, doc=p.doc, tf=p.tf, doc_len=p.doc_len, idf=p.idf,
                     avgdl=np.float64(p.avgdl), k1=np.float64(p.k1), b=np.float64(p.b), vocab=vocab)
        token_len = 0
        if tokens is not None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32)
            assert tokens.shape[0] == n and token_lens is not None
            np.save(os.path.join(self.path, "tokens.npy"), tokens)
            np.save(os.path.join(self.path, "token_lens.npy"), np.asarray(token_lens, dtype=np.int32))
            token_len = int(tokens.shape[1])
        with open(os.path.join(self.path, "meta.json"), "w") as f:
            json.dump({"version": FORMAT_VERSION, "n_rows": n, "dim": self.dim, "tenants": self._tenant_id,
                       "has_bm25": bool(build_bm25 and n), "token_len": token_len}, f)
        return self.path


def export_table(rows: Iterable, path: str, dim: int = 1536, build_bm25: bool = True) -> str:
    """rows: what `SELECT id, agent_id, content, embedding, metadata, created_at FROM document_chunks|archival_memory`
    yields (tuples in that column order, or dicts with those keys)."""
    w = ShardWriter(path, dim)
    for r in rows:
        if isinstance(r, dict):
            w.add(r["id"], r["agent_id"], r["content"], r["embedding"], r.get("metadata"), r.get("created_at"),
                  r.get("filename"), r.get("file_type"))
        else:
            w.add(*r[:6])
    return w.close(build_bm25=build_bm25)


class Shard:
    """Read side: memory-mapped arrays + lazy payload access."""

    def __init__(self, path: str):
        self.path = path
        with open(os.path.join(path, "meta.json")) as f:
            self.meta = json.load(f)
        if self.meta["version"] != FORMAT_VERSION:
            raise ValueError("unsupported shard format version %r" % (self.meta["version"],))
        self.n_rows, self.dim = int(self.meta["n_rows"]), int(self.meta["dim"])
        self.tenant_table: Dict[str, int] = dict(self.meta["tenants"])
        self.embeddings = np.load(os.path.join(path, "embeddings.npy"), mmap_mode="r")
        self.ids = np.load(os.path.join(path, "ids.npy"), mmap_mode="r")
        self.tenants = np.load(os.path.join(path, "tenants.npy"), mmap_mode="r")
        self.created_at = np.load(os.path.join(path, "created_at.npy"), mmap_mode="r")
        self._idx = np.load(os.path.join(path, "rows.idx.npy"), mmap_mode="r")
        self._rows = open(os.path.join(path, "rows.jsonl"), "rb")
        self.tokens = self.token_lens = None
        if self.meta.get("token_len"):
            self.tokens = np.load(os.path.join(path, "tokens.npy"), mmap_mode="r")
            self.token_lens = np.load(os.path.join(path, "token_lens.npy"), mmap_mode="r")

    def row(self, i: int) -> Dict[str, Any]:
        self._rows.seek(int(self._idx[i]))
        return json.loads(self._rows.read(int(self._idx[i + 1] - self._idx[i])).decode("utf-8"))

    def postings(self) -> Optional[Bm25Postings]:
        if not self.meta.get("has_bm25"):
            return None
        z = np.load(os.path.join(self.path, "bm25.npz"), allow_pickle=True)
        vocab = {w: i for i, w in enumerate(z["vocab"].tolist())}
        return Bm25Postings(z["indptr"], z["doc"], z["tf"], z["doc_len"], z["idf"], float(z["avgdl"]), vocab,
                            float(z["k1"]), float(z["b"]))

    def temporal_scores(self, now: datetime, recency_weight: float = 0.15, half_life_days: float = 30.0) -> np.ndarray:
        """RECENCY_WEIGHT * 0.5 ** (days_old / half_life) per row, 0.0 where created_at is NULL
        (/root/reference/rag/retrieval.py:266-292; `now` naive = UTC, as the stored timestamps)."""
        t = _epoch(now)
        days = (t - np.asarray(self.created_at)) / 86400.0
        out = recency_weight * np.power(0.5, days / half_life_days)
        return np.where(np.isnan(days), 0.0, out)

    def close(self):
        self._rows.close()


def open_shard(path: str) -> Shard:
    return Shard(path)


def load_shard_into(engine, shard: Shard, begin: int = 0, end: Optional[int] = None, chunk_rows: int = 131072,
                    with_bm25: bool = True):
    """Stream rows [begin, end) of the shard into `engine` (RagEngine): fp32 master + fp16 unit copy are built on the
    device chunk by chunk; doc ids = the table's primary keys; tenant filter and (optionally) the BM25 slice loaded.
    Returns the Bm25Postings that were loaded (or None)."""
    end = shard.n_rows if end is None else end
    n = end - begin
    engine.index_reserve(max(n, 1))
    for b in range(begin, end, chunk_rows):
        engine.index_append(np.ascontiguousarray(shard.embeddings[b:min(end, b + chunk_rows)]))
    if n:
        engine.set_ids(np.asarray(shard.ids[begin:end]))
        engine.set_tenants(np.asarray(shard.tenants[begin:end]))
    post = None
    if with_bm25 and n:
        post = shard.postings()
        if post is not None:
            if begin != 0 or end != shard.n_rows:
                post = post.shard(begin, end)
            post.load(engine)
    return post
