"""the dispatch class that drives the two re-rankers. The golden file was produced by running the
REFERENCE SelectiveReranker with recording fakes (tools/make_golden_selective.py); the mirror must make the same calls
(which re-ranker, positional `(query, results, top_k)`, `is_available()` probes in the same order), return the same rows
and keep the same counters, for every combination of available re-rankers, intents (enum and string) and score shapes.
CPU-only: no arithmetic is involved."""
import json
import os

from optimized_rag_amd.selective_reranker import QueryIntent, SelectiveReranker


class Fake:
    def __init__(self, name, log, available=True):
        self.name, self.log, self.available = name, log, available

    def is_available(self):
        self.log.append([self.name, "is_available"])
        return self.available

    def rerank(self, *args, **kwargs):
        self.log.append([self.name, "rerank", len(args), sorted(kwargs)])
        results, top_k = args[1], (args[2] if len(args) > 2 else kwargs.get("top_k"))
        return list(reversed(results))[:top_k]


def test_dispatch_matches_the_reference_on_every_scenario(golden_dir):
    with open(os.path.join(golden_dir, "selective_reranker.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 600
    for c in cases:
        log = []
        o = Fake("openai", log) if c["have_openai"] else None
        x = Fake("cross", log, bool(c["cross_available"])) if c["have_cross"] else None
        sr = SelectiveReranker(openai_reranker=o, cross_encoder_reranker=x, enable_selective=c["enable_selective"])
        intent = QueryIntent[c["intent"]] if c["intent_kind"] == "enum" else c["intent"]
        res = [{"content": f"d{i}", "score": s, "pos": i} for i, s in enumerate(c["scores"])]
        out = sr.rerank("the query", res, intent=intent, top_k=c["top_k"])
        assert log == c["calls"], c
        assert [d["pos"] for d in out] == c["returned_pos"], c
        assert sr.get_statistics() == c["stats"], c


def test_mirror_rerankers_accept_the_dispatchers_call_shape():
    """The dispatcher calls `rerank(query, results, top_k)` positionally and probes `is_available()`: the mirror re-rankers
    take exactly that (signatures only; their arithmetic is covered by the GPU tests)."""
    import inspect
    from optimized_rag_amd.reranker import CrossEncoderReranker, OpenAIReranker
    for cls in (CrossEncoderReranker, OpenAIReranker):
        params = list(inspect.signature(cls.rerank).parameters)
        assert params[:4] == ["self", "query", "results", "top_k"], (cls, params)
    assert callable(CrossEncoderReranker.is_available)
