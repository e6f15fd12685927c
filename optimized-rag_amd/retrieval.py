"""HybridRetriever — same class/method surface as /root/reference/rag/retrieval.py:13-371, arithmetic on MI355X.

What runs where
  * cosine(query, every doc)      -> rag_pairwise_cosine_host (float64 kernel)   [reference :253-256, :362-371]
  * BM25Okapi.get_scores + /max   -> rag_bm25_scores_adhoc_host (stateless)      [reference :324-347]
  * alpha*s + beta*kw + gamma*t, stable sort, [:top_k] -> rag_linear_fuse_topk_host [reference :294-322]
  * tokenising, keyword-set overlap fallback, ISO timestamps, dict assembly stay in Python (text handling).
`retrieve`/`_retrieve_*` are the reference's fan-out wrappers; with a GpuDocumentIndex as `document_store` the
`ORDER BY embedding <=> q LIMIT k` behind them is the HIP dense top-k.
"""
import logging
from datetime import datetime
from typing import Any, Dict, List, Optional

from .bm25 import Bm25Postings
from .engine import as_matrix, get_engine

logger = logging.getLogger(__name__)

ENABLE_TEMPORAL_BOOST = True     # reference config.py:38
RECENCY_WEIGHT = 0.15            # reference config.py:39
RECENCY_HALF_LIFE_DAYS = 30      # reference config.py:40


class HybridRetriever:
    INTENT_WEIGHTS = {           # reference retrieval.py:22-47
        'question_answering': {'alpha': 0.55, 'beta': 0.40, 'gamma': 0.05},
        'fact_checking': {'alpha': 0.50, 'beta': 0.45, 'gamma': 0.05},
        'multi_hop_reasoning': {'alpha': 0.60, 'beta': 0.30, 'gamma': 0.10},
        'comparison': {'alpha': 0.50, 'beta': 0.45, 'gamma': 0.05},
        'summarization': {'alpha': 0.65, 'beta': 0.25, 'gamma': 0.10},
        'search': {'alpha': 0.45, 'beta': 0.50, 'gamma': 0.05},
        'clarification': {'alpha': 0.70, 'beta': 0.20, 'gamma': 0.10},
        'conversational': {'alpha': 0.70, 'beta': 0.20, 'gamma': 0.10},
        'default': {'alpha': 0.55, 'beta': 0.35, 'gamma': 0.10},
    }

    def __init__(self, memory_manager, document_store, agent_id: str, alpha: float = 0.55, beta: float = 0.35,
                 gamma: float = 0.10, weight_manager=None, use_adaptive_weights: bool = True, *, engine=None):
        self.memory_manager = memory_manager
        self.document_store = document_store
        self.agent_id = agent_id
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.weight_manager = weight_manager           # stored and unused, as in the reference (:79)
        self.use_adaptive_weights = use_adaptive_weights
        self._engine = engine
        # BM25 is native here (the reference needs the optional rank_bm25 package, :113-120)
        self.bm25_available = True
        logger.info("HybridRetriever (MI355X): adaptive_weights=%s default_weights=(%.2f, %.2f, %.2f)",
                    use_adaptive_weights, alpha, beta, gamma)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def get_weights_for_intent(self, intent: str) -> tuple:
        intent_key = intent.lower().replace(' ', '_') if intent else 'default'
        w = self.INTENT_WEIGHTS.get(intent_key, self.INTENT_WEIGHTS['default'])
        return w['alpha'], w['beta'], w['gamma']

    # ---- multi-source fan-out (reference :122-212) --------------------------------------------------
    def retrieve(self, query: str, sources: List[str], top_k: int = 20) -> List[Dict[str, Any]]:
        all_results = []
        if 'archival' in sources or 'archival_memory' in sources:
            all_results.extend(self._retrieve_archival(query, top_k))
        if 'documents' in sources:
            all_results.extend(self._retrieve_documents(query, top_k))
        if 'conversation' in sources or 'conversation_history' in sources:
            all_results.extend(self._retrieve_conversation(query, top_k))
        logger.info("Retrieved %d total results from %d sources", len(all_results), len(sources))
        return all_results

    def retrieve_batch(self, queries: List[str], sources: List[str], top_k: int = 20) -> List[List[Dict[str, Any]]]:
        """Batched `retrieve`: element i equals `retrieve(queries[i], sources, top_k)`. The
        `documents` source - the pgvector scan behind hierarchical_retriever.py:454-458 - is answered for ALL queries by one
        GPU call when the document store offers `search_many` (GpuDocumentIndex does); the other sources keep the
        reference's per-query calls (they are remote / SQL round trips, not arithmetic)."""
        docs = None
        if 'documents' in sources and hasattr(self.document_store, 'search_many'):
            try:
                docs = self.document_store.search_many(self.agent_id, list(queries), top_k=top_k)
                for res in docs:
                    for r in res:
                        r['source'] = 'documents'
            except Exception as e:
                logger.error("Document retrieval failed: %s", e)
                docs = [[] for _ in queries]
        out = []
        for i, query in enumerate(queries):
            res = []
            if 'archival' in sources or 'archival_memory' in sources:
                res.extend(self._retrieve_archival(query, top_k))
            if 'documents' in sources:
                res.extend(docs[i] if docs is not None else self._retrieve_documents(query, top_k))
            if 'conversation' in sources or 'conversation_history' in sources:
                res.extend(self._retrieve_conversation(query, top_k))
            out.append(res)
        logger.info("Retrieved results for %d queries from %d sources", len(queries), len(sources))
        return out

    def _retrieve_archival(self, query: str, top_k: int) -> List[Dict[str, Any]]:
        try:
            results = self.memory_manager.archival_memory_search(query, top_k=top_k)
            for r in results:
                r['source'] = 'archival_memory'
            return results
        except Exception as e:
            logger.error("Archival retrieval failed: %s", e)
            return []

    def _retrieve_documents(self, query: str, top_k: int) -> List[Dict[str, Any]]:
        try:
            results = self.document_store.search(agent_id=self.agent_id, query=query, top_k=top_k)
            for r in results:
                r['source'] = 'documents'
            return results
        except Exception as e:
            logger.error("Document retrieval failed: %s", e)
            return []

    def _retrieve_conversation(self, query: str, top_k: int) -> List[Dict[str, Any]]:
        try:
            conversation_id = self.memory_manager.agent_id
            results = self.memory_manager.conversation_search(conversation_id, query, limit=top_k)
            return [{'content': m['content'], 'source': 'conversation_history',
                     'metadata': {'role': m['role'], 'timestamp': m.get('created_at', '')}, 'similarity': 0.5}
                    for m in results]
        except Exception as e:
            logger.error("Conversation retrieval failed: %s", e)
            return []

    # ---- hybrid_search (reference :214-322); no try/except there either --------------------------
    def _now(self):
        return datetime.now()

    def _temporal_scores(self, n, documents_metadata):
        if not (documents_metadata and ENABLE_TEMPORAL_BOOST):
            return [0.0] * n
        now = self._now()
        out = []
        for md in documents_metadata:
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
                    val = RECENCY_WEIGHT * (0.5 ** (days_old / RECENCY_HALF_LIFE_DAYS))
            out.append(val)
        return out

    def hybrid_search(self, query: str, corpus: List[str], embeddings: List[List[float]],
                      query_embedding: List[float], top_k: int = 10,
                      documents_metadata: Optional[List[Dict[str, Any]]] = None,
                      query_intent: Optional[str] = None) -> List[Dict[str, Any]]:
        if self.use_adaptive_weights and query_intent:
            alpha, beta, gamma = self.get_weights_for_intent(query_intent)
        else:
            alpha, beta, gamma = self.alpha, self.beta, self.gamma
        n = len(corpus)
        semantic = self._semantic_scores(query_embedding, embeddings)
        keyword = self._bm25_scores(query, corpus) if self.bm25_available else self._simple_keyword_scores(query, corpus)
        temporal = self._temporal_scores(n, documents_metadata)
        if n == 0:
            return []
        idx, hybrid = self.engine.linear_fuse_topk(semantic, keyword, temporal, alpha, beta, gamma, max(0, min(top_k, n)))
        ranked = []
        for i in idx:
            i = int(i)
            r = {'content': corpus[i], 'hybrid_score': float(hybrid[i]), 'semantic_score': float(semantic[i]),
                 'keyword_score': float(keyword[i]), 'temporal_score': float(temporal[i]), 'embedding': embeddings[i]}
            if documents_metadata and i < len(documents_metadata):
                r['metadata'] = documents_metadata[i]
            ranked.append(r)
        return ranked

    def _semantic_scores(self, query_embedding, embeddings) -> List[float]:
        if len(embeddings) == 0:
            return []
        m = as_matrix([query_embedding] + list(embeddings))
        return self.engine.pairwise_cosine(m[:1], m[1:])[0].tolist()

    def _bm25_scores(self, query: str, corpus: List[str]) -> List[float]:
        if not corpus or all(len(d.split()) == 0 for d in corpus):
            logger.warning("BM25: Empty or whitespace-only corpus, returning zeros")
            return [0.0] * len(corpus)
        # stateless: a fresh BM25Okapi per call in the reference (:333-341); the index's resident postings stay loaded
        post = Bm25Postings.from_corpus(corpus)
        ptr, terms = post.encode_queries([query])
        scores = self.engine.bm25_scores_adhoc(post.indptr, post.doc, post.tf, post.doc_len, post.idf, post.avgdl, ptr, terms,
                                               post.k1, post.b)[0]
        mx = float(scores.max()) if len(scores) > 0 and scores.max() > 0 else 1.0
        return [float(s / mx) for s in scores]

    def _simple_keyword_scores(self, query: str, corpus: List[str]) -> List[float]:
        q_terms = set(query.lower().split())
        return [(len(q_terms & set(d.lower().split())) / len(q_terms)) if q_terms else 0.0 for d in corpus]

    def _cosine_similarity(self, vec1: List[float], vec2: List[float]) -> float:
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])
