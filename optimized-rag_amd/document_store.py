"""GpuDocumentIndex — the HBM-resident replacement of the pgvector tables, with `DocumentStore.search`'s surface
(/root/reference/rag/document_store.py:424-485) and `search_archival_memory`'s (database/operations.py:110-159).

`ORDER BY dc.embedding <=> %s::vector LIMIT %s` with `WHERE dc.agent_id = %s` becomes rag_dense_topk_* with a
per-row tenant filter; the exact-scan result (not the HNSW approximation) is what is reproduced. Row payloads
(content, filename, metadata) stay on the host; only embeddings live in HBM."""
import logging
from typing import Any, Dict, List, Optional

import numpy as np

from .engine import get_engine

logger = logging.getLogger(__name__)


class _ShardRows:
    """List-like view of a shard's payload rows [begin, end): `rows[i]` reads one JSON line from disk."""

    def __init__(self, shard, begin, end):
        self.shard, self.begin, self.end = shard, begin, end

    def __len__(self):
        return self.end - self.begin

    def __getitem__(self, i):
        g = self.begin + int(i)
        r = self.shard.row(g)
        r["id"] = int(self.shard.ids[g])
        ts = float(self.shard.created_at[g])
        r["created_at"] = None if ts != ts else ts
        return r


class GpuDocumentIndex:
    def __init__(self, embedding_service, dim: int = 1536, *, engine=None):
        self.embeddings = embedding_service            # same attribute name the reference's DocumentStore uses
        self.dim = dim
        self._engine = engine
        self.rows: List[Dict[str, Any]] = []
        self._tenant_id: Dict[str, int] = {}

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine(self.dim)
        return self._engine

    def bulk_load(self, rows: List[Dict[str, Any]], embeddings) -> None:
        """rows[i]: {content, agent_id, filename?, file_type?, metadata?, id?, created_at?}; embeddings [N, dim] float32
        (the export of document_chunks / archival_memory)."""
        emb = np.ascontiguousarray(embeddings, dtype=np.float32)
        assert emb.shape == (len(rows), self.dim)
        self.rows = list(rows)
        tenants = np.empty(len(rows), dtype=np.int32)
        for i, r in enumerate(rows):
            tenants[i] = self._tenant_id.setdefault(str(r.get("agent_id", "")), len(self._tenant_id))
        self.engine.index_load(emb)
        self.engine.set_tenants(tenants)

    def load_shard(self, shard, begin: int = 0, end: Optional[int] = None, chunk_rows: int = 131072) -> None:
        """Stream an exported shard directory (shard_format.py; path or open Shard) into the index: rows [begin, end),
        payloads stay on disk and are read lazily per hit, doc ids are the table's primary keys."""
        from . import shard_format as SF
        sh = SF.open_shard(shard) if isinstance(shard, str) else shard
        end = sh.n_rows if end is None else end
        assert sh.dim == self.dim
        SF.load_shard_into(self.engine, sh, begin, end, chunk_rows)
        self._tenant_id = dict(sh.tenant_table)
        self.rows = _ShardRows(sh, begin, end)

    def _search_rows(self, agent_id, query_embeddings, top_k):
        if agent_id is not None and str(agent_id) not in self._tenant_id:
            Q = np.asarray(query_embeddings).reshape(-1, self.dim).shape[0]
            return np.full((Q, top_k), -1, dtype=np.int32), np.zeros((Q, top_k))
        tenant = -1 if agent_id is None else self._tenant_id[str(agent_id)]
        _, rows, scores = self.engine.dense_topk(np.asarray(query_embeddings, dtype=np.float32).reshape(-1, self.dim),
                                                 top_k, tenant=tenant)
        return rows, scores

    def search(self, agent_id: str, query: str, top_k: int = 5, with_embeddings: bool = True) -> List[Dict[str, Any]]:
        try:
            q = self.embeddings.generate_embedding(query)
            rows, scores = self._search_rows(agent_id, [q], top_k)
            hit = [int(r) for r in rows[0] if r >= 0]
            embs = self.engine.fetch_rows(hit) if (with_embeddings and hit) else None
            out = []
            for j, r in enumerate(hit):
                row = self.rows[r]
                d = {"content": row.get("content", ""), "filename": row.get("filename"), "file_type": row.get("file_type"),
                     "score": float(scores[0][j]), "metadata": row.get("metadata") or {}}
                if embs is not None:
                    d["embedding"] = embs[j].tolist()                 # Python floats; saves apply_mmr's re-embedding calls
                out.append(d)
            return out
        except Exception as e:                                       # reference: log and return [] (:483-485)
            logger.error("Search failed: %s", e)
            return []

    def search_many(self, agent_id: str, queries: List[str], top_k: int = 5, with_embeddings: bool = True) -> List[List[Dict[str, Any]]]:
        """Batched `search` (the reference agent can only submit one query per call, so nothing in
        its surface reaches the batched throughput): ONE embedding call for all queries when the service offers
        `generate_embeddings_batch`, ONE rag_dense_topk_host call with Q = len(queries), one row fetch. Element i equals
        `search(agent_id, queries[i], top_k, with_embeddings)`; on failure every element is [] (the reference's log-and-
        return-empty, :483-485)."""
        try:
            if not queries:
                return []
            if hasattr(self.embeddings, "generate_embeddings_batch"):
                embs = self.embeddings.generate_embeddings_batch(list(queries))
            else:
                embs = [self.embeddings.generate_embedding(q) for q in queries]
            rows, scores = self._search_rows(agent_id, embs, top_k)
            hits = [[int(r) for r in rows[i] if r >= 0] for i in range(len(queries))]
            flat = [r for h in hits for r in h]
            fetched = self.engine.fetch_rows(flat) if (with_embeddings and flat) else None
            out, pos = [], 0
            for i, h in enumerate(hits):
                res = []
                for j, r in enumerate(h):
                    row = self.rows[r]
                    d = {"content": row.get("content", ""), "filename": row.get("filename"), "file_type": row.get("file_type"),
                         "score": float(scores[i][j]), "metadata": row.get("metadata") or {}}
                    if fetched is not None:
                        d["embedding"] = fetched[pos + j].tolist()
                    res.append(d)
                pos += len(h)
                out.append(res)
            return out
        except Exception as e:
            logger.error("Batched search failed: %s", e)
            return [[] for _ in queries]

    def search_batch(self, agent_id: Optional[str], query_embeddings, top_k: int = 20):
        """Batched entry the reference lacks: Q query embeddings at once -> (row indices [Q,k], cosines [Q,k])."""
        return self._search_rows(agent_id, query_embeddings, top_k)

    def search_archival_memory(self, agent_id: str, query_embedding: List[float], limit: int = 5) -> List[Dict[str, Any]]:
        rows, scores = self._search_rows(agent_id, [query_embedding], limit)
        return [{"id": self.rows[int(r)].get("id", int(r)), "content": self.rows[int(r)].get("content", ""),
                 "metadata": self.rows[int(r)].get("metadata"), "similarity": float(s),
                 "created_at": self.rows[int(r)].get("created_at")}
                for r, s in zip(rows[0], scores[0]) if r >= 0]
