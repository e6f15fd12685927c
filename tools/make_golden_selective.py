#!/usr/bin/env python3
"""Golden vectors for SURVEY.md section 8 row a11: the REFERENCE SelectiveReranker (/root/reference/rag/selective_reranker.py,
loaded by path) is driven with recording fake re-rankers over a grid of scenarios; what it called (which re-ranker, with
which positional / keyword arguments), what it returned and its counters are written to tests/golden/selective_reranker.json.
Data only - no reference source is copied. Run here (the GPU box has no /root/reference): python tools/make_golden_selective.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402

MG._install_stubs()
ia = MG._load_by_path("rag.models.intent_analysis", "rag/models/intent_analysis.py")
sys.modules["rag.models.intent_analysis"] = ia
sys.modules["rag"].QueryIntent = ia.QueryIntent            # `from rag import QueryIntent` inside _should_rerank
ref = MG._load_by_path("_ref_selective", "rag/selective_reranker.py")


class Fake:
    def __init__(self, name, log, available=True):
        self.name, self.log, self.available = name, log, available

    def is_available(self):
        self.log.append([self.name, "is_available"])
        return self.available

    def rerank(self, *args, **kwargs):
        self.log.append([self.name, "rerank", len(args), sorted(kwargs)])
        results, top_k = args[1], (args[2] if len(args) > 2 else kwargs.get("top_k"))
        return list(reversed(results))[:top_k]             # a recognisable permutation


def results_of(scores):
    return [{"content": f"d{i}", "score": s, "pos": i} for i, s in enumerate(scores)]


SCORE_SETS = {"few_low": [0.01, 0.02, 0.03], "few_ok": [0.5, 0.4, 0.3, 0.2], "flat": [0.5] * 12,
              "spread": [0.0, 1.0] * 5 + [0.5, 0.5], "mid": [0.1, 0.9, 0.1, 0.5, 0.5, 0.5, 0.5, 0.5, 0.1, 0.9], "none": []}
INTENTS = [("enum", m.name) for m in ia.QueryIntent] + [("str", s) for s in ("qa", "chat", "multi_hop", "compare", "factual", "other")]
cases = []
for enable in (False, True):
    for have_o, have_c, c_avail in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 0)):
        for kind, iv in INTENTS:
            for sname in (["flat"] if not enable else list(SCORE_SETS)):
                log = []
                o = Fake("openai", log) if have_o else None
                c = Fake("cross", log, bool(c_avail)) if have_c else None
                sr = ref.SelectiveReranker(openai_reranker=o, cross_encoder_reranker=c, enable_selective=enable)
                intent = ia.QueryIntent[iv] if kind == "enum" else iv
                res = results_of(SCORE_SETS[sname])
                out = sr.rerank("the query", res, intent=intent, top_k=4)
                cases.append({"enable_selective": enable, "have_openai": have_o, "have_cross": have_c, "cross_available": c_avail,
                              "intent_kind": kind, "intent": iv, "scores": SCORE_SETS[sname], "top_k": 4,
                              "calls": log, "returned_pos": [d["pos"] for d in out], "stats": sr.get_statistics()})
with open(os.path.join(MG.OUT, "selective_reranker.json"), "w") as f:
    json.dump({"cases": cases}, f)
print(len(cases), "cases")
