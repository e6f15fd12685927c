"""Batched Tier-2 entry for the hierarchical retriever.

The reference's `HierarchicalRetriever._retrieve_tier_2` (/root/reference/rag/hierarchical_retriever.py:440-468) asks
`hybrid_retriever.retrieve(query=, sources=["documents"], top_k=)` for ONE query, tags every row with `tier: 2` and returns
[] on any error. `retrieve_tier_2_batch` is the same step for a list of queries, answered by one GPU search through
`HybridRetriever.retrieve_batch`; the tiers' confidence / escalation logic (LLM calls, :30-106, :222-367) stays the
reference's own."""
import logging
from typing import Any, Dict, List

logger = logging.getLogger(__name__)


def retrieve_tier_2_batch(hybrid_retriever, queries: List[str], top_k: int) -> List[List[Dict[str, Any]]]:
    try:
        batches = hybrid_retriever.retrieve_batch(list(queries), sources=["documents"], top_k=top_k)
        for results in batches:
            for r in results:
                r['tier'] = 2
        return batches
    except Exception as e:
        logger.error("Tier 2 retrieval error: %s", e)
        return [[] for _ in queries]
