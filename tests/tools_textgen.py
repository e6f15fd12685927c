"""Synthetic text helpers shared by tests (same generator family as tools/make_golden.py)."""
WORDS = ("system memory vector index query document retrieval ranking fusion agent graph node "
         "embedding cosine score keyword search context token model latency cache batch shard "
         "kernel bandwidth matrix tile stream buffer policy storage engine network protocol "
         "database table column record update delete insert commit branch merge release "
         "Paris London Berlin Madrid Rome Lisbon Vienna Prague Dublin Oslo").split()


def make_sentence(rng, n_lo=6, n_hi=14, end="."):
    n = int(rng.integers(n_lo, n_hi))
    w = [WORDS[int(i)] for i in rng.integers(0, len(WORDS), n)]
    w[0] = w[0].capitalize()
    return " ".join(w) + end


def make_doc(rng, n_sent):
    return " ".join(make_sentence(rng) for _ in range(n_sent))
