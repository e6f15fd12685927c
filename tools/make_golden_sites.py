#!/usr/bin/env python3
"""Golden vectors for the "other cosine call sites" (SURVEY.md §8f.3), produced by RUNNING the reference's own code in
this container: SemanticChunker.chunk (rag/chunking.py:140-220), Deduplicator.semantic_dedup
(rag/data_wrangler.py:294-326), EnsembleVerifier._embedding_verification (rag/ensemble_verifier.py:237-272),
ClaimAlignmentScorer._semantic_similarity (rag/claim_alignment.py:284-318) and
ConversationReferenceDetector._detect_semantic_reference (rag/conversation_reference_detector.py:108-198).

Only data is written (tests/golden/cosine_sites.json): the text -> embedding table the fake embedding service served,
the inputs and what the reference returned. The five modules have stdlib-only imports and are loaded by file path.

Run once, here (the GPU box has no /root/reference):   python tools/make_golden_sites.py
"""
import dataclasses
import hashlib
import importlib.util
import json
import os

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cosine_sites.json")
DIM = 48


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class TableService:
    """Deterministic embeddings: a topic direction chosen by the text's first word + hash-seeded noise, rounded to
    float32 so that the device (float32 rows) and the reference (Python floats) see identical values."""

    def __init__(self, noise):
        self.table, self.noise = {}, noise
        self.topics = np.random.default_rng(7).standard_normal((6, DIM))

    def generate_embedding(self, text):
        if text not in self.table:
            h = int(hashlib.md5(text.encode()).hexdigest()[:8], 16)
            rng = np.random.default_rng(h)
            first = (text.split() or ["x"])[0].lower()
            topic = self.topics[int(hashlib.md5(first.encode()).hexdigest()[:4], 16) % 6]
            v = topic + self.noise * rng.standard_normal(DIM)
            if "ZEROVEC" in text:
                v = np.zeros(DIM)
            self.table[text] = [float(x) for x in v.astype(np.float32)]
        return self.table[text]

    def generate_embeddings_batch(self, texts):
        return [self.generate_embedding(t) for t in texts]


def main():
    chunking = load("_ref_chunking", "rag/chunking.py")
    wrangler = load("_ref_wrangler", "rag/data_wrangler.py")
    verifier = load("_ref_verifier", "rag/ensemble_verifier.py")
    align = load("_ref_align", "rag/claim_alignment.py")
    conv = load("_ref_conv", "rag/conversation_reference_detector.py")
    svc = TableService(noise=0.9)
    out = {"dim": DIM}

    # ---- SemanticChunker.chunk ---------------------------------------------------------------------------------
    rng = np.random.default_rng(11)
    openers = ["Alpha", "Beta", "Gamma", "Delta", "Omega", "Sigma"]

    def text_of(n_sent, run):
        sents, op = [], openers[0]
        for i in range(n_sent):
            if i % run == 0:
                op = openers[int(rng.integers(0, 6))]
            body = " ".join(f"tok{int(x)}" for x in rng.integers(0, 50, int(rng.integers(4, 14))))
            sents.append(f"{op} {body}{'.!?'[i % 3]}")
        return " ".join(sents)

    cases = []
    for (n_sent, run, thr, mx, mn) in [(14, 3, 0.7, 1500, 200), (25, 4, 0.55, 400, 120), (9, 2, 0.9, 300, 40),
                                        (6, 1, 0.3, 5000, 10), (1, 1, 0.7, 1500, 200), (12, 3, 0.7, 150, 60)]:
        text = text_of(n_sent, run)
        ch = chunking.SemanticChunker(svc, similarity_threshold=thr, max_chunk_size=mx, min_chunk_size=mn)
        cases.append({"text": text, "threshold": thr, "max_chunk_size": mx, "min_chunk_size": mn,
                      "metadata": {"source": "golden"}, "expected": ch.chunk(text, {"source": "golden"})})
    short = "Alpha tiny."
    ch = chunking.SemanticChunker(svc)
    cases.append({"text": short, "threshold": 0.7, "max_chunk_size": 1500, "min_chunk_size": 200, "metadata": None,
                  "expected": ch.chunk(short)})
    cases.append({"text": "", "threshold": 0.7, "max_chunk_size": 1500, "min_chunk_size": 200, "metadata": None,
                  "expected": ch.chunk("")})
    out["chunker"] = cases

    # ---- Deduplicator.semantic_dedup ---------------------------------------------------------------------------
    svc_tight = TableService(noise=0.25)
    dd = []
    for thr in (0.95, 0.8, 0.5):
        texts = [f"{openers[int(rng.integers(0, 6))]} item {i}" for i in range(30)] + ["Alpha ZEROVEC"]
        texts[7] = texts[2]                                          # exact duplicate text -> identical embedding
        embs = [svc_tight.generate_embedding(t) for t in texts]
        chunks = [{"content": t, "i": i} for i, t in enumerate(texts)]
        kept = wrangler.Deduplicator.semantic_dedup(chunks, embs, threshold=thr)
        dd.append({"texts": texts, "threshold": thr, "expected_kept": [c["i"] for c in kept]})
    out["dedup"] = dd

    # ---- EnsembleVerifier._embedding_verification --------------------------------------------------------------
    ev = verifier.EnsembleVerifier(llm=None, embedding_service=svc_tight)
    vcases = []
    for claim, docs in [("Alpha claim about tokens", ["Alpha doc one", "Beta doc two", "", "Gamma doc three"]),
                        ("Beta another claim", ["Gamma only", "Delta only"]),
                        ("Omega lonely", []),
                        ("Sigma zero", ["Sigma ZEROVEC"])]:
        documents = [{"content": d} for d in docs]
        vcases.append({"claim": claim, "docs": docs, "threshold": ev.embedding_threshold,
                       "expected": ev._embedding_verification(claim, documents)})
    out["verifier"] = vcases

    # ---- ClaimAlignmentScorer._semantic_similarity -------------------------------------------------------------
    sc = align.ClaimAlignmentScorer(embedding_service=svc_tight)
    acases = []
    for claim, doc in [("Alpha the claim under test", "Beta sentence number one is long enough. Alpha sentence number two is "
                        "also long enough! Gamma third sentence of the document here? short. Alpha another matching sentence."),
                       ("Delta claim", "tiny. small. no."),
                       ("Omega claim", " ".join(f"Omega sentence number {i} with enough characters." for i in range(26)))]:
        s, best = sc._semantic_similarity(claim, doc)
        acases.append({"claim": claim, "document": doc, "expected_score": s, "expected_sentence": best})
    out["alignment"] = acases

    # ---- ConversationReferenceDetector._detect_semantic_reference ----------------------------------------------
    det = conv.ConversationReferenceDetector(llm=None, embedding_service=svc_tight, semantic_threshold=0.75)
    ccases = []
    for query, msgs in [("Alpha what about it", [{"content": "Beta first message"}, {"content": "Alpha second message"},
                                                   {"content": ""}, "Alpha plain string message"]),
                        ("Gamma a much longer query with more than ten words in it for sure yes", [{"content": "Gamma topic"},
                                                                                                   {"content": "Delta topic"}]),
                        ("Sigma nothing", []),
                        ("Delta unrelated", [{"content": "Omega far away"}])]:
        r = det._detect_semantic_reference(query, msgs)
        ccases.append({"query": query, "messages": msgs, "expected": dataclasses.asdict(r)})
    out["conversation"] = ccases

    out["embeddings"] = {**svc.table, **svc_tight.table}
    out["embeddings_noise09"] = svc.table
    out["embeddings_noise025"] = svc_tight.table
    del out["embeddings"]
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
