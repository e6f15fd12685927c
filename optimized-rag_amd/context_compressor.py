"""ContextCompressor with the reference's surface (/root/reference/rag/context_compressor.py:17-371).
The query x sentence cosine loop (:227-228) runs on the GPU (rag_pairwise_cosine_host); sentence splitting,
lexical overlap, early-outs and dict assembly are text handling and stay in Python."""
import logging
import re
from typing import Any, Dict, List, Tuple

from .engine import as_matrix, get_engine

logger = logging.getLogger(__name__)

COMPRESSION_MIN_THRESHOLD = 0.005                               # reference config.py:215
COMPRESSION_INTENT_THRESHOLDS = {"QUESTION_ANSWERING": 0.25, "SEARCH": 0.2, "CONVERSATIONAL": 0.15,
                                 "MULTI_HOP_REASONING": 0.3}    # reference config.py:216-221
_STOP = {'the', 'a', 'an', 'and', 'or', 'but', 'in', 'on', 'at', 'to', 'for', 'of', 'with', 'by', 'from', 'is', 'was',
         'are', 'were', 'be', 'been', 'being'}


class ContextCompressor:
    def __init__(self, max_tokens: int = 4000, sentences_per_doc: int = 8, embedding_service=None,
                 conservative_mode: bool = True, *, engine=None):
        self.max_tokens = max_tokens
        self.sentences_per_doc = sentences_per_doc
        self.embedding_service = embedding_service
        self.conservative_mode = conservative_mode
        self.use_semantic_scoring = embedding_service is not None
        self.semantic_weight, self.lexical_weight = 0.7, 0.3
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def compress(self, query: str, documents: List[Dict[str, Any]], max_tokens=None, query_intent="question_answering",
                 confidence: float = 1.0) -> List[Dict[str, Any]]:
        if not documents:
            return []
        if len(documents) <= 7:
            return documents
        if self.conservative_mode and confidence >= 0.6:
            total_chars = sum(len(d.get('content', '')) for d in documents)
            if total_chars <= (max_tokens or self.max_tokens) * 4:
                return documents
        if confidence < 0.6:
            return self._concatenate_without_compression(documents, max_tokens or self.max_tokens)
        if confidence < 0.8:
            sentences_per_doc, mult = self.sentences_per_doc + 3, 0.6
        else:
            sentences_per_doc, mult = self.sentences_per_doc, 1.0
        # the reference looks the lower-case enum value up in UPPER-case keys, so the 0.45 default always wins
        intent_key = query_intent.value if hasattr(query_intent, 'value') else str(query_intent)
        base_threshold = COMPRESSION_INTENT_THRESHOLDS.get(intent_key, 0.45) * mult
        if len(documents) <= 5:
            max_doc_score = max((d.get('score', 0) for d in documents), default=0)
            relevance_threshold = COMPRESSION_MIN_THRESHOLD if max_doc_score < 0.5 else base_threshold
        else:
            relevance_threshold = base_threshold
        documents = [d for d in documents if d.get('score', 1.0) >= relevance_threshold]
        if not documents:
            logger.warning("All documents below relevance threshold (%s), returning empty context", relevance_threshold)
            return []
        compressed = []
        for doc in documents:
            content = doc.get('content', '')
            sentences = self._split_sentences(content)
            if not sentences:
                continue
            if self.use_semantic_scoring:
                scored = self._score_sentences_hybrid(query, sentences)
            else:
                scored = [(s, self._score_sentence_lexical(query, s)) for s in sentences]
            scored.sort(key=lambda x: x[1], reverse=True)
            top = set(s for s, _ in scored[:sentences_per_doc])
            ordered = [s for s in sentences if s in top]
            text = ' '.join(ordered)
            compressed.append({**doc, 'content': text, 'original_content': content, 'compressed': True,
                               'original_length': len(content), 'compressed_length': len(text),
                               'compression_ratio': len(text) / len(content) if len(content) > 0 else 0,
                               'sentences_kept': len(ordered), 'sentences_total': len(sentences)})
        return compressed

    def _split_sentences(self, text: str) -> List[str]:
        if not text:
            return []
        return [s.strip() for s in re.split(r'[.!?]+\s+', text) if len(s.strip()) > 20]

    def _score_sentences_hybrid(self, query: str, sentences: List[str]) -> List[Tuple[str, float]]:
        try:
            if not self.embedding_service:
                raise ValueError("Embedding service not available")
            q_emb = self.embedding_service.generate_embedding(query)
            s_embs = self.embedding_service.generate_embeddings_batch(sentences)
            m = as_matrix([q_emb] + list(s_embs))
            sims = self.engine.pairwise_cosine(m[:1], m[1:])[0]           # 1 x S on the GPU
            return [(s, self.semantic_weight * float(c) + self.lexical_weight * self._score_sentence_lexical(query, s))
                    for s, c in zip(sentences, sims)]
        except Exception as e:
            logger.error("Semantic scoring failed, falling back to lexical: %s", e)
            return [(s, self._score_sentence_lexical(query, s)) for s in sentences]

    def _cosine_similarity(self, vec1, vec2) -> float:
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])

    def _score_sentence_lexical(self, query: str, sentence: str) -> float:
        ql, sl = query.lower(), sentence.lower()
        qw = set(re.findall(r'\b\w+\b', ql)) - _STOP
        sw = set(re.findall(r'\b\w+\b', sl)) - _STOP
        if not qw:
            return 0.0
        score = len(qw & sw) / len(qw)
        if ql in sl:
            score += 0.2
        return min(score, 1.0)

    def _concatenate_without_compression(self, documents, max_tokens: int):
        result, total = [], 0
        for doc in documents:
            content = doc.get('content', '')
            if total + len(content) <= max_tokens:
                result.append({**doc, 'compressed': False, 'preservation_reason': 'low_confidence_skip_compression'})
                total += len(content)
            else:
                remaining = max_tokens - total
                if remaining > 200:
                    result.append({**doc, 'content': content[:remaining], 'compressed': False, 'truncated': True,
                                   'original_length': len(content), 'truncated_length': remaining})
                break
        return result

    def get_compression_stats(self, compressed_docs) -> Dict[str, Any]:
        if not compressed_docs:
            return {'total_documents': 0, 'total_original_length': 0, 'total_compressed_length': 0, 'tokens_saved': 0,
                    'compression_ratio': 0, 'avg_sentences_kept': 0}
        orig = sum(d.get('original_length', 0) for d in compressed_docs)
        comp = sum(d.get('compressed_length', 0) for d in compressed_docs)
        return {'total_documents': len(compressed_docs), 'total_original_length': orig, 'total_compressed_length': comp,
                'tokens_saved': orig - comp, 'compression_ratio': comp / orig if orig > 0 else 0,
                'avg_sentences_kept': sum(d.get('sentences_kept', 0) for d in compressed_docs) / len(compressed_docs)}
