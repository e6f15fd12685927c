"""SelectiveReranker — dispatch between the two re-rankers, same surface as /root/reference/rag/selective_reranker.py:14-260.

Pure control flow: nothing here touches the GPU. It exists so that the object graph of
`MemGPTRAGAgent._initialize_rag` can be built entirely from this package: it drives `CrossEncoderReranker.rerank(query,
results, top_k)` / `.is_available()` and `OpenAIReranker.rerank(query, results, top_k)` (positional, as the reference
calls them, :205-226). Behaviour is pinned by tests/golden/selective_reranker.json, produced by running the reference
class with recording fakes (tools/make_golden_selective.py)."""
import logging
from enum import Enum
from typing import Any, Dict, List

logger = logging.getLogger(__name__)


class QueryIntent(Enum):                      # values of rag/models/intent_analysis.py:6-15
    QUESTION_ANSWERING = "question_answering"
    SUMMARIZATION = "summarization"
    COMPARISON = "comparison"
    FACT_CHECKING = "fact_checking"
    MULTI_HOP_REASONING = "multi_hop_reasoning"
    CLARIFICATION = "clarification"
    CONVERSATIONAL = "conversational"
    INSTRUCTION = "instruction"
    SEARCH = "search"


def _value(intent):
    return intent.value if hasattr(intent, "value") else str(intent).lower()


_PRECISION = {"qa", "multi_hop", "compare", "factual", "question_answering", "comparison", "fact_checking", "summarization",
              "search"}
_PRECISION_ENUM = {"question_answering", "multi_hop_reasoning", "comparison", "fact_checking", "summarization", "search"}
_FACTUAL = {"qa", "multi_hop", "compare", "question_answering", "multi_hop_reasoning", "comparison", "fact_checking"}
_CONVERSATIONAL = {"chat", "search", "conversational", "clarification"}


class SelectiveReranker:
    def __init__(self, openai_reranker=None, cross_encoder_reranker=None, enable_selective: bool = False):
        self.openai_reranker = openai_reranker
        self.cross_encoder_reranker = cross_encoder_reranker
        self.enable_selective = enable_selective
        self.total_queries = 0
        self.reranking_skipped = 0
        self.reranking_applied = 0

    def rerank(self, query: str, results: List[Dict[str, Any]], intent=QueryIntent.QUESTION_ANSWERING,
               top_k: int = 5) -> List[Dict[str, Any]]:
        self.total_queries += 1
        if not self.enable_selective:             # default: always rerank (the counters stay untouched, as in the reference)
            return self._apply_reranking(query, results, intent, top_k)
        go, reason = self._should_rerank(results, intent)
        if not go:
            self.reranking_skipped += 1
            logger.info("Skipping reranking: %s", reason)
            return results[:top_k]
        self.reranking_applied += 1
        return self._apply_reranking(query, results, intent, top_k)

    def _should_rerank(self, results, intent):
        v = _value(intent)
        # an enum member counts through its own value; a plain string through the string table (:104-118)
        if (hasattr(intent, "value") and v in _PRECISION_ENUM) or v in _PRECISION:
            return True, f"Precision intent ({v}) - always rerank"
        if len(results) <= 5:
            scores = [r.get("score", 0) for r in results]
            avg = sum(scores) / len(scores) if scores else 0
            if avg < 0.05:
                return True, f"Low embedding scores ({avg:.3f}), CrossEncoder needed"
            return False, "Too few results (≤5)"
        scores = [r.get("score", 0) for r in results[:10]]
        if not scores:
            return True, "No scores available"
        avg = sum(scores) / len(scores)
        var = sum((s - avg) ** 2 for s in scores) / len(scores)
        if var > 0.1:
            return False, f"High score variance ({var:.3f})"
        if var < 0.05:
            return True, f"Low score variance ({var:.3f})"
        if intent in ["qa", "multi_hop", "compare"]:
            return True, f"Intent requires precision ({intent})"
        if intent == "chat":
            if scores[0] < 0.7:
                return True, "Low top score for chat"
            return False, "Chat query with good top score"
        return True, "Default policy"

    def _cross_ok(self):
        return bool(self.cross_encoder_reranker and self.cross_encoder_reranker.is_available())

    def _apply_reranking(self, query, results, intent, top_k):
        v = _value(intent)
        if v in _FACTUAL:
            if self._cross_ok():
                return self.cross_encoder_reranker.rerank(query, results, top_k)
            if self.openai_reranker:
                return self.openai_reranker.rerank(query, results, top_k)
        elif v in _CONVERSATIONAL:
            if self.openai_reranker:
                return self.openai_reranker.rerank(query, results, top_k)
            if self._cross_ok():
                return self.cross_encoder_reranker.rerank(query, results, top_k)
        if self._cross_ok():
            return self.cross_encoder_reranker.rerank(query, results, top_k)
        if self.openai_reranker:
            return self.openai_reranker.rerank(query, results, top_k)
        logger.warning("No reranker available for intent: %s, returning original results", v)
        return results[:top_k]

    def get_statistics(self) -> Dict[str, Any]:
        skip = self.reranking_skipped / self.total_queries if self.total_queries > 0 else 0
        return {"total_queries": self.total_queries, "reranking_applied": self.reranking_applied,
                "reranking_skipped": self.reranking_skipped, "skip_rate": skip, "skip_rate_percent": f"{skip * 100:.1f}%",
                "estimated_cost_savings": f"{skip * 100:.0f}% of reranking costs"}
