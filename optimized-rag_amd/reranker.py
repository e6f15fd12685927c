"""Re-rankers and fusion with the reference's surface (/root/reference/rag/reranker.py), arithmetic on MI355X.

  OpenAIReranker.rerank        0.7*cos + 0.3*original (:28-90)      cosine -> rag_pairwise_cosine_host
  MMRDiversifier.diversify     greedy MMR (:116-195)                one (n+1)x(n+1) cosine matrix on the GPU
  ReciprocalRankFusion.fuse    sum 1/(k+rank), content-keyed (:224-271) -> rag_rrf_fuse_host
  CrossEncoderReranker.rerank  BERT cross-encoder logits + sigmoid (:320-384) -> rag_ce_score_host
Inputs are lists of dicts that are mutated in place exactly as the reference does; errors are logged and answered
with the reference's documented fallbacks (`results[:top_k]`), never raised into the agent graph.
"""
import logging
import math
import os
from typing import Any, Dict, List

import numpy as np

from .engine import as_matrix, get_engine

logger = logging.getLogger(__name__)


class _EngineMixin:
    _engine = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine


class OpenAIReranker(_EngineMixin):
    def __init__(self, openai_client, model: str = "text-embedding-3-large", *, engine=None):
        self.client = openai_client
        self.model = model
        self._engine = engine

    def rerank(self, query: str, results: List[Dict[str, Any]], top_k: int = 5) -> List[Dict[str, Any]]:
        if not results:
            return []
        try:
            contents = [query] + [r.get('content', '')[:8000] for r in results]
            resp = self.client.embeddings.create(input=contents, model=self.model)      # remote API, out of scope
            embs = [resp.data[i].embedding for i in range(len(results) + 1)]
            m = as_matrix(embs)
            sims = self.engine.pairwise_cosine(m[:1], m[1:])[0]
            for i, result in enumerate(results):
                original = result.get('similarity', 0) or result.get('score', 0)
                result['rerank_score'] = 0.7 * float(sims[i]) + 0.3 * original
                result['embedding'] = embs[i + 1]
            return sorted(results, key=lambda x: x['rerank_score'], reverse=True)[:top_k]
        except Exception as e:
            logger.error("OpenAI reranking failed: %s", e)
            return results[:top_k]

    def _cosine_similarity(self, vec1, vec2) -> float:
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])


def _valid_embedding(e):
    return bool(e) and isinstance(e, list) and len(e) > 0 and \
        all(isinstance(v, (int, float)) and not math.isnan(v) and not math.isinf(v) for v in e)


class MMRDiversifier(_EngineMixin):
    def __init__(self, lambda_param: float = 0.7, *, engine=None):
        self.lambda_param = lambda_param
        self._engine = engine

    def diversify(self, query_embedding: List[float], results: List[Dict[str, Any]], top_k: int = 5):
        if not results:
            return []
        valid = [r for r in results if _valid_embedding(r.get('embedding'))]
        if not valid:
            logger.warning("MMR: No valid embeddings found, returning original results")
            return results[:top_k]
        m = as_matrix([query_embedding if query_embedding else []] + [r['embedding'] for r in valid])
        lam = self.lambda_param
        if len(valid) <= self.engine.MMR_MAX_CANDIDATES:
            # the whole greedy loop runs on the device (rag_mmr_select_host, variant 0)
            selected, scores = self.engine.mmr_select(m[0], m[1:], top_k, lam, 0)
            for i, sc in zip(selected, scores):
                valid[int(i)]['mmr_score'] = float(sc)
            return [valid[int(i)] for i in selected]
        # larger pools: one cosine matrix [query | docs] x [docs] on the device, greedy loop on the host
        S = self.engine.pairwise_cosine(m, m[1:])
        rel, sim = S[0], S[1:]
        selected, remaining = [], list(range(len(valid)))
        while len(selected) < top_k and remaining:
            best, best_s = None, None
            for i in remaining:
                diversity = 1 - max(sim[i, s] for s in selected) if selected else 1.0
                score = lam * float(rel[i]) + (1 - lam) * float(diversity)
                if best is None or score > best_s:           # max() keeps the first maximal element
                    best, best_s = i, score
            valid[best]['mmr_score'] = best_s
            selected.append(best)
            remaining.remove(best)
        return [valid[i] for i in selected]

    def _cosine_similarity(self, vec1, vec2) -> float:
        if not vec1 or not vec2:
            return 0.0
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])


class ReciprocalRankFusion(_EngineMixin):
    def __init__(self, k: int = 60, *, engine=None):
        self.k = k
        self._engine = engine

    def fuse(self, result_lists: List[List[Dict[str, Any]]], top_k: int = 10) -> List[Dict[str, Any]]:
        # the reference keys on the `content` string (:245): number the distinct contents, fuse ids on the GPU
        key_of, doc_of = {}, []
        L = len(result_lists)
        width = max([len(l) for l in result_lists] + [1])
        lists = np.full((1, max(L, 1), width), -1, dtype=np.int64)
        for li, lst in enumerate(result_lists):
            for j, doc in enumerate(lst):
                c = doc.get('content', '')
                if c not in key_of:
                    key_of[c] = len(doc_of)
                    doc_of.append(doc)
                lists[0, li, j] = key_of[c]
        if not doc_of or top_k <= 0:
            return []
        keys, scores, _ = self.engine.rrf_fuse(lists, rrf_k=self.k, top_k=min(top_k, len(doc_of)))
        fused = []
        for key, s in zip(keys[0], scores[0]):
            if key < 0:
                break
            doc = doc_of[int(key)]
            doc['rrf_score'] = float(s)
            fused.append(doc)
        return fused


class CrossEncoderReranker(_EngineMixin):
    """`model_name` is a LOCAL directory holding config.json, model.safetensors and vocab.txt of a
    BertForSequenceClassification cross-encoder (e.g. a downloaded cross-encoder/ms-marco-MiniLM-L-6-v2). The
    reference hands the same argument to sentence_transformers.CrossEncoder (:312-313), which fetches by name;
    there is no network here, so a name that is not a directory leaves the model unavailable — the same
    outcome as the reference's swallowed load failure (:315-318)."""

    TENSOR_ORDER_DOC = "see cross_encoder.flatten_state_dict"

    def __init__(self, model_name: str = "cross-encoder/ms-marco-MiniLM-L-6-v2", max_length: int = 512, *, engine=None):
        self.model_name = model_name
        self.max_length = max_length
        self.model = None
        self._engine = engine
        try:
            from .cross_encoder import LocalCrossEncoder
            if os.path.isdir(model_name):
                self.model = LocalCrossEncoder.from_dir(model_name, max_length=max_length, engine=self.engine)
                logger.info("Initialized MI355X CrossEncoder from %s", model_name)
            else:
                logger.error("CrossEncoder model directory %r not found (no network: models load from local paths)", model_name)
        except Exception as e:
            logger.error("Failed to load CrossEncoder model: %s", e, exc_info=True)

    def rerank(self, query: str, results: List[Dict[str, Any]], top_k: int = 5) -> List[Dict[str, Any]]:
        if not results:
            return []
        if self.model is None:
            logger.warning("CrossEncoder not available, returning original results")
            return results[:top_k]
        try:
            pairs = []
            for result in results:
                content = result.get('content', '')
                if len(content) > 2000:
                    content = content[:2000]
                pairs.append([query, content])
            scores = self.model.predict(pairs)                                  # raw logits, GPU
            normalized = [1 / (1 + math.exp(-s)) for s in scores]
            for result, score, norm in zip(results, scores, normalized):
                if 'score' in result and 'embedding_score' not in result:
                    result['embedding_score'] = result['score']
                result['score'] = float(norm)
                result['cross_encoder_score'] = float(norm)
                result['cross_encoder_raw_score'] = float(score)
            return sorted(results, key=lambda x: x['cross_encoder_score'], reverse=True)[:top_k]
        except Exception as e:
            logger.error("CrossEncoder reranking failed: %s", e)
            return results[:top_k]

    def is_available(self) -> bool:
        return self.model is not None
