"""ConsistencyChecker with the reference's surface (/root/reference/rag/consistency_checker.py:15-281).
The all-pairs claim cosine loop (:169-189) becomes ONE C x C float64 cosine matrix on the GPU
(rag_pairwise_cosine_host); claim extraction and the negation/number heuristics are text handling and stay here."""
import logging
import re
from typing import Any, Dict, List

import numpy as np

from .engine import as_matrix, get_engine

logger = logging.getLogger(__name__)

_META_PATTERNS = [r'^(this|that|these|those|it|they)\s+(is|are|was|were)', r'^(here|there)\s+(is|are)',
                  r'^(in conclusion|in summary|overall|finally)']
_NEGATION_PAIRS = [("is not", "is"), ("are not", "are"), ("was not", "was"), ("were not", "were"),
                   ("does not", "does"), ("do not", "do"), ("did not", "did"), ("cannot", "can"),
                   ("will not", "will"), ("should not", "should"), ("no", "yes"), ("false", "true"),
                   ("incorrect", "correct"), ("never", "always")]


class ConsistencyChecker:
    def __init__(self, embedding_service, similarity_threshold: float = 0.85, *, engine=None):
        self.embedding_service = embedding_service
        self.similarity_threshold = similarity_threshold
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def check_consistency(self, documents: List[Dict[str, Any]], query: str) -> Dict[str, Any]:
        if len(documents) < 2:
            return {"consistent": True, "contradictions": [], "confidence": 1.0, "warning": None}
        try:
            all_claims = []
            for idx, doc in enumerate(documents):
                for claim in self._extract_claims(doc.get("content", "")):
                    all_claims.append({"text": claim, "doc_idx": idx, "source": doc.get("source", f"doc_{idx}")})
            if len(all_claims) < 2:
                return {"consistent": True, "contradictions": [], "confidence": 1.0,
                        "warning": "Too few claims to check consistency"}
            contradictions = self._find_contradictions(all_claims)
            total_pairs = len(all_claims) * (len(all_claims) - 1) / 2
            score = 1.0 - min(len(contradictions) / max(total_pairs, 1), 1.0)
            if contradictions:
                logger.warning("Consistency check found %d contradictions (score: %.2f)", len(contradictions), score)
            return {"consistent": len(contradictions) == 0 or score >= 0.8, "contradictions": contradictions[:5],
                    "contradiction_count": len(contradictions), "confidence": score, "total_claims": len(all_claims),
                    "warning": self._generate_warning(contradictions) if contradictions else None}
        except Exception as e:                                   # fail open, as the reference does (:105-112)
            logger.error("Consistency check failed: %s", e)
            return {"consistent": True, "contradictions": [], "confidence": 0.5,
                    "warning": f"Consistency check error: {str(e)}"}

    def _extract_claims(self, text: str) -> List[str]:
        claims = []
        for sent in re.split(r'[.!?]+', text):
            sent = sent.strip()
            if len(sent) < 20 or any(re.match(p, sent.lower()) for p in _META_PATTERNS):
                continue
            claims.append(sent)
        return claims

    def _find_contradictions(self, claims: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        try:
            embeddings = self.embedding_service.generate_embeddings_batch([c["text"] for c in claims])
        except Exception as e:
            logger.error("Failed to compute embeddings: %s", e)
            return []
        m = as_matrix(embeddings)
        S = self.engine.pairwise_cosine(m)                      # C x C, float64, one launch
        out = []
        # the reference walks all i < j pairs in Python (:169-189); same pairs in the same order, but the same-document and
        # threshold tests run over the whole upper triangle at once and only the survivors reach the string checks
        doc = np.asarray([c["doc_idx"] for c in claims])
        iu, ju = np.triu_indices(len(claims), 1)
        keep = (doc[iu] != doc[ju]) & (S[iu, ju] >= self.similarity_threshold)
        for i, j in zip(iu[keep].tolist(), ju[keep].tolist()):
            sim = float(S[i, j])
            if self._is_contradiction(claims[i]["text"], claims[j]["text"]):
                out.append({"claim_1": claims[i]["text"][:200], "claim_2": claims[j]["text"][:200],
                            "source_1": claims[i]["source"], "source_2": claims[j]["source"],
                            "similarity": round(sim, 3), "type": "semantic_contradiction"})
        return out

    def _is_contradiction(self, text1: str, text2: str) -> bool:
        a, b = text1.lower(), text2.lower()
        for neg, pos in _NEGATION_PAIRS:
            if (neg in a and pos in b) or (pos in a and neg in b):
                return True
        n1 = re.findall(r'\b\d+\.?\d*\b', text1)
        n2 = re.findall(r'\b\d+\.?\d*\b', text2)
        return bool(n1 and n2 and set(n1) != set(n2))

    def _cosine_similarity(self, vec1, vec2) -> float:
        m = as_matrix([vec1, vec2])
        return float(self.engine.pairwise_cosine(m[:1], m[1:])[0, 0])

    def _generate_warning(self, contradictions) -> str:
        n = len(contradictions)
        if n == 1:
            return "Warning: Found 1 potential contradiction in sources. Response may be unreliable."
        if n <= 3:
            return f"Warning: Found {n} contradictions in sources. Please verify information."
        return f"Warning: Found {n} contradictions in sources. High uncertainty in response."
