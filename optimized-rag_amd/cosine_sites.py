"""The remaining pairwise-cosine call sites of the reference, on the device.

Each function / class keeps the reference's name, arguments, return shape and swallowed-error behaviour; the Python
generator-expression cosines become one call into librag_hip.so. Text handling (regex sentence splitting, dict
assembly, caches) stays on the host exactly as the reference does it.

  SemanticChunker.chunk                                   /root/reference/rag/chunking.py:140-239
  Deduplicator.semantic_dedup                             /root/reference/rag/data_wrangler.py:294-326
  EnsembleVerifier._embedding_verification                /root/reference/rag/ensemble_verifier.py:237-272
  ClaimAlignmentScorer._semantic_similarity               /root/reference/rag/claim_alignment.py:284-318
  ConversationReferenceDetector._detect_semantic_reference /root/reference/rag/conversation_reference_detector.py:108-198
"""
import logging
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

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


# ---------------------------------------------------------------------------------------------------------------
class SemanticChunker(_EngineMixin):
    """Semantic chunking based on embedding similarity (rag/chunking.py:140-239). The sentence loop — a cosine against a
    running pairwise average, so sequential by construction — runs as one kernel (rag_chunk_chain_host)."""

    def __init__(self, embedding_service, similarity_threshold: float = 0.7, max_chunk_size: int = 1500,
                 min_chunk_size: int = 200, *, engine=None):
        self.embedding_service = embedding_service
        self.similarity_threshold = similarity_threshold
        self.max_chunk_size = max_chunk_size
        self.min_chunk_size = min_chunk_size
        self._engine = engine

    def chunk(self, text: str, metadata: Optional[Dict] = None) -> List[Dict[str, Any]]:
        sentences = self._split_sentences(text)
        if len(sentences) == 0:
            return []
        if len(text) < self.min_chunk_size:
            return [{"content": text, "metadata": {**(metadata or {}), "chunk_id": 0}}]
        embeddings = self.embedding_service.generate_embeddings_batch(sentences)
        groups = self.engine.chunk_chain(as_matrix(embeddings), [len(s) for s in sentences], self.similarity_threshold,
                                         self.max_chunk_size, self.min_chunk_size)
        chunks, start = [], 0
        for i in range(1, len(sentences) + 1):
            if i == len(sentences) or groups[i] != groups[start]:
                chunks.append(self._create_chunk(sentences[start:i], int(groups[start]), metadata))
                start = i
        logger.info("Created %d semantic chunks", len(chunks))
        return chunks

    def _split_sentences(self, text: str) -> List[str]:
        sentences = re.split(r'(?<=[.!?])\s+', text)
        return [s.strip() for s in sentences if s.strip()]

    def _cosine_similarity(self, vec1: List[float], vec2: List[float]) -> float:
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])

    def _create_chunk(self, sentences: List[str], chunk_id: int, base_metadata: Optional[Dict]) -> Dict[str, Any]:
        content = " ".join(sentences)
        return {"content": content, "metadata": {"chunk_id": chunk_id, "num_sentences": len(sentences),
                                                 "chunk_size": len(content), **(base_metadata or {})}}


# ---------------------------------------------------------------------------------------------------------------
class Deduplicator:
    @staticmethod
    def semantic_dedup(chunks: List[Dict[str, Any]], embeddings: List[List[float]], threshold: float = 0.95, *,
                       engine=None) -> List[Dict[str, Any]]:
        """Remove semantically similar chunks (rag/data_wrangler.py:294-326): a chunk is kept unless its cosine with an
        already kept chunk reaches `threshold`. One n x n cosine matrix on the device, then the greedy scan over booleans."""
        n = min(len(chunks), len(embeddings))                      # zip() semantics
        if n == 0:
            return []
        m = as_matrix(embeddings[:n])
        dup = (engine or get_engine()).pairwise_cosine(m, m) >= threshold
        kept: List[int] = []
        for i in range(n):
            if not (kept and dup[i, kept].any()):
                kept.append(i)
        logger.info("Semantic dedup: %d → %d", len(chunks), len(kept))
        return [chunks[i] for i in kept]


# ---------------------------------------------------------------------------------------------------------------
def embedding_verification(claim_embedding: List[float], doc_embeddings: List[List[float]], embedding_threshold: float = 0.60,
                           *, engine=None) -> Dict[str, Any]:
    """Core of EnsembleVerifier._embedding_verification (rag/ensemble_verifier.py:237-272): best similarity of the claim
    against the (non-empty) documents, starting from 0.0; supported iff it exceeds the threshold."""
    try:
        best = 0.0
        if doc_embeddings:
            m = as_matrix([claim_embedding] + list(doc_embeddings))
            sims = (engine or get_engine()).pairwise_cosine(m[:1], m[1:])[0]
            best = max(0.0, float(sims.max()))
        return {"supported": best > embedding_threshold, "confidence": best, "method": "embedding"}
    except Exception as e:
        logger.error("Embedding verification failed: %s", e)
        return {"supported": False, "confidence": 0.0, "method": "embedding"}


class EmbeddingVerifierMixin(_EngineMixin):
    """Drop-in for the embedding leg of EnsembleVerifier: same method name, reads `self.embedding_service` /
    `self.embedding_threshold` like the reference; documents are truncated to 2000 characters and empty ones skipped."""

    embedding_threshold = 0.60

    def _embedding_verification(self, claim: str, documents: List[Dict[str, Any]]) -> Dict[str, Any]:
        try:
            contents = [d.get("content", "")[:2000] for d in documents if d.get("content", "")]
            claim_emb = self.embedding_service.generate_embedding(claim)
            doc_embs = [self.embedding_service.generate_embedding(c) for c in contents]
            return embedding_verification(claim_emb, doc_embs, self.embedding_threshold, engine=self.engine)
        except Exception as e:
            logger.error("Embedding verification failed: %s", e)
            return {"supported": False, "confidence": 0.0, "method": "embedding"}


# ---------------------------------------------------------------------------------------------------------------
def semantic_similarity(claim: str, document_content: str, embedding_service, *, engine=None) -> Tuple[float, str]:
    """ClaimAlignmentScorer._semantic_similarity (rag/claim_alignment.py:284-318): best cosine of the claim against the
    first 20 sentences (> 20 characters) of the document; strict `>` from 0.0 keeps the FIRST best sentence."""
    try:
        sentences = [s.strip() for s in re.split(r'[.!?]+', document_content) if len(s.strip()) > 20]
        if not sentences:
            return 0.0, ''
        sentences = sentences[:20]
        claim_emb = embedding_service.generate_embedding(claim)
        embs = [embedding_service.generate_embedding(s) for s in sentences]
        m = as_matrix([claim_emb] + embs)
        sims = (engine or get_engine()).pairwise_cosine(m[:1], m[1:])[0]
        best_score, best_sentence = 0.0, ''
        for s, sim in zip(sentences, sims):
            if sim > best_score:
                best_score, best_sentence = float(sim), s
        return best_score, best_sentence[:200]
    except Exception as e:
        logger.warning("Semantic similarity failed: %s", e)
        return 0.0, ''


# ---------------------------------------------------------------------------------------------------------------
@dataclass
class ConversationReferenceResult:
    is_conversation_reference: bool
    confidence: float
    method: str
    reasoning: str
    referenced_message_index: Optional[int] = None


class SemanticReferenceDetector(_EngineMixin):
    """The semantic leg of ConversationReferenceDetector (rag/conversation_reference_detector.py:108-198)."""

    def __init__(self, embedding_service, semantic_threshold: float = 0.75, *, engine=None):
        self.embedding_service = embedding_service
        self.semantic_threshold = semantic_threshold
        self._history_embeddings_cache: Dict[int, List[float]] = {}
        self._engine = engine

    def _detect_semantic_reference(self, query: str, messages: List[Dict]) -> ConversationReferenceResult:
        try:
            query_embedding = self.embedding_service.generate_embedding(query)
            history = []
            for i, msg in enumerate(messages):
                content = msg.get('content', '') if isinstance(msg, dict) else str(msg)
                if not content:
                    continue
                key = hash(content[:100])
                if key not in self._history_embeddings_cache:
                    self._history_embeddings_cache[key] = self.embedding_service.generate_embedding(content[:500])
                history.append((i, self._history_embeddings_cache[key]))
            if not history:
                return ConversationReferenceResult(False, 0.0, 'semantic', 'No valid history messages')
            m = as_matrix([query_embedding] + [e for _, e in history])
            sims = self.engine.pairwise_cosine(m[:1], m[1:])[0]
            j = int(np.argmax(sims))                        # stable sort desc, first element == first maximum
            best_idx, best_sim = history[j][0], float(sims[j])
            if len(query.split()) <= 10 and best_sim > self.semantic_threshold:
                return ConversationReferenceResult(True, best_sim, 'semantic',
                                                   f'Query semantically similar to message #{best_idx} (sim={best_sim:.2f})', best_idx)
            elif best_sim > 0.85:
                return ConversationReferenceResult(True, best_sim, 'semantic', f'Strong semantic match with message #{best_idx}',
                                                   best_idx)
            return ConversationReferenceResult(False, best_sim, 'semantic', f'Similarity ({best_sim:.2f}) below threshold')
        except Exception as e:
            logger.error("Semantic detection failed: %s", e)
            return ConversationReferenceResult(False, 0.0, 'semantic', f'Error: {e}')
