"""apply_mmr / cosine_similarity with the reference's surface (/root/reference/rag/nodes/helpers.py:183-290).
One (n+1) x n cosine matrix on the GPU replaces the O(k*n*|selected|) Python cosine calls. When the documents
come from GpuDocumentIndex.search they already carry their `embedding`, so the reference's per-document
re-embedding HTTP calls (:215-223, ~7 s per query in its own log) never happen."""
import logging
from typing import Any, Dict, List

from .engine import as_matrix, get_engine

logger = logging.getLogger(__name__)


def apply_mmr(query: str, documents: List[Dict[str, Any]], lambda_: float, k: int, embedding_service, *, engine=None):
    if len(documents) <= k:
        return documents
    try:
        eng = engine or get_engine()
        q_emb = embedding_service.generate_embedding(query)
        embs = []
        for doc in documents:
            if "embedding" in doc and doc["embedding"]:
                embs.append(doc["embedding"])
            else:
                emb = embedding_service.generate_embedding(doc.get("content", doc.get("text", "")))
                doc["embedding"] = emb
                embs.append(emb)
        m = as_matrix([q_emb] + embs)
        if len(documents) <= eng.MMR_MAX_CANDIDATES:           # greedy loop on the device (rag_mmr_select_host, variant 1)
            selected, _ = eng.mmr_select(m[0], m[1:], k, lambda_, 1)
            return [documents[int(i)] for i in selected]
        S = eng.pairwise_cosine(m, m[1:])
        rel, sim = S[0], S[1:]
        selected, remaining = [], list(range(len(documents)))
        while len(selected) < k and remaining:
            best, best_s = None, None
            for i in remaining:
                max_sim = max(sim[i, s] for s in selected) if selected else 0.0
                mmr = lambda_ * float(rel[i]) - (1 - lambda_) * float(max_sim)
                if best is None or mmr > best_s:
                    best, best_s = i, mmr
            selected.append(best)
            remaining.remove(best)
        return [documents[i] for i in selected]
    except Exception as e:
        logger.error("MMR calculation failed: %s", e, exc_info=True)
        return documents[:k]


def cosine_similarity(vec1: List[float], vec2: List[float], *, engine=None) -> float:
    try:
        m = as_matrix([vec1, vec2])
        return float((engine or get_engine()).pairwise_cosine(m[:1], m[1:])[0, 0])
    except Exception as e:
        logger.error("Cosine similarity calculation failed: %s", e, exc_info=True)
        return 0.0
