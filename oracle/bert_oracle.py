"""TEST INFRASTRUCTURE — CPU oracle for the cross-encoder forward (K7). Not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates: `CrossEncoderReranker.rerank` calls `self.model.predict(pairs)` on
`sentence_transformers.CrossEncoder("cross-encoder/ms-marco-MiniLM-L-6-v2")`
(/root/reference/rag/reranker.py:312-313,355). sentence-transformers (>=2.2.0, requirements.txt:24)
and the checkpoint are NOT in the reference tree or in this image, so this restates the published
BertForSequenceClassification forward (post-LN BERT, erf-GELU, LayerNorm eps 1e-12, pooler tanh,
1-logit classifier) in numpy float64.

Pinning: checked against the `transformers` BertForSequenceClassification shipped in this image
(tests/test_oracle_bert.py, and tests/golden/bert_minilm.npz produced by tools/make_golden.py).
Real-checkpoint parity is UNPINNED (no weights offline): weights are seeded random.
"""
import math

import numpy as np


def minilm_config():
    """Shape of cross-encoder/ms-marco-MiniLM-L-6-v2 (public model card; reference names it at config.py:49)."""
    return dict(vocab_size=30522, hidden=384, layers=6, heads=12, ffn=1536, max_pos=512, type_vocab=2, eps=1e-12)


def seeded_weights(cfg, seed):
    """Deterministic float32 state-dict (HF BertForSequenceClassification key names, nn.Linear [out,in])."""
    rng = np.random.default_rng(seed)
    H, F = cfg["hidden"], cfg["ffn"]

    def mat(*shape, s=0.05):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    w = {}
    w["bert.embeddings.word_embeddings.weight"] = mat(cfg["vocab_size"], H, s=0.1)
    w["bert.embeddings.position_embeddings.weight"] = mat(cfg["max_pos"], H, s=0.1)
    w["bert.embeddings.token_type_embeddings.weight"] = mat(cfg["type_vocab"], H, s=0.1)
    w["bert.embeddings.LayerNorm.weight"] = (1.0 + mat(H, s=0.1)).astype(np.float32)
    w["bert.embeddings.LayerNorm.bias"] = mat(H, s=0.1)
    for l in range(cfg["layers"]):
        p = f"bert.encoder.layer.{l}."
        for nm in ("query", "key", "value"):
            w[p + f"attention.self.{nm}.weight"] = mat(H, H, s=0.08)
            w[p + f"attention.self.{nm}.bias"] = mat(H, s=0.05)
        w[p + "attention.output.dense.weight"] = mat(H, H)
        w[p + "attention.output.dense.bias"] = mat(H)
        w[p + "attention.output.LayerNorm.weight"] = (1.0 + mat(H, s=0.1)).astype(np.float32)
        w[p + "attention.output.LayerNorm.bias"] = mat(H, s=0.1)
        w[p + "intermediate.dense.weight"] = mat(F, H)
        w[p + "intermediate.dense.bias"] = mat(F)
        w[p + "output.dense.weight"] = mat(H, F)
        w[p + "output.dense.bias"] = mat(H)
        w[p + "output.LayerNorm.weight"] = (1.0 + mat(H, s=0.1)).astype(np.float32)
        w[p + "output.LayerNorm.bias"] = mat(H, s=0.1)
    w["bert.pooler.dense.weight"] = mat(H, H)
    w["bert.pooler.dense.bias"] = mat(H)
    w["classifier.weight"] = mat(1, H, s=0.5)
    w["classifier.bias"] = mat(1, s=0.5)
    return w


_erf = np.vectorize(math.erf, otypes=[np.float64])


def _ln(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def forward_logits(w, cfg, input_ids, token_type_ids, lens, dtype=np.float64, fast_erf=False):
    """[P,L] ids, [P] valid lengths -> [P] raw logits (what CrossEncoder.predict returns for this checkpoint).
    fast_erf: scipy.special.erf instead of the per-element math.erf (same function to ~1 ulp; tests/test_oracle_bert.py
    bounds the difference) so that hundreds of 256-token pairs finish in seconds."""
    W, x = forward_hidden(w, cfg, input_ids, token_type_ids, lens, dtype, fast_erf)
    pooled = np.tanh(x[:, 0] @ W["bert.pooler.dense.weight"].T + W["bert.pooler.dense.bias"])
    return (pooled @ W["classifier.weight"].T + W["classifier.bias"])[:, 0]


def sentence_embeddings(w, cfg, input_ids, token_type_ids, lens, normalize=True, dtype=np.float64, fast_erf=False):
    """sentence-transformers' Pooling(mean) + Normalize over the BERT encoder: mean of the last hidden state over the real
    tokens, then x / max(|x|, 1e-12) (torch.nn.functional.normalize). The pooler / classifier weights are not used."""
    _, x = forward_hidden(w, cfg, input_ids, token_type_ids, lens, dtype, fast_erf)
    ok = (np.arange(x.shape[1])[None, :] < np.asarray(lens)[:, None]).astype(dtype)[:, :, None]
    pooled = (x * ok).sum(1) / np.maximum(ok.sum(1), 1e-9)
    if normalize:
        pooled = pooled / np.maximum(np.linalg.norm(pooled, axis=1, keepdims=True), 1e-12)
    return pooled


def forward_hidden(w, cfg, input_ids, token_type_ids, lens, dtype=np.float64, fast_erf=False):
    """The encoder: (weights as `dtype`, last hidden state [P, L, H])."""
    erf = _erf
    if fast_erf:
        from scipy.special import erf
    W = {k: v.astype(dtype) for k, v in w.items()}
    P, L = input_ids.shape
    H, nh = cfg["hidden"], cfg["heads"]
    dh = H // nh
    x = (W["bert.embeddings.word_embeddings.weight"][input_ids]
         + W["bert.embeddings.token_type_embeddings.weight"][token_type_ids]
         + W["bert.embeddings.position_embeddings.weight"][np.arange(L)][None])
    x = _ln(x, W["bert.embeddings.LayerNorm.weight"], W["bert.embeddings.LayerNorm.bias"], cfg["eps"])
    key_ok = np.arange(L)[None, :] < np.asarray(lens)[:, None]           # [P,L]
    add_mask = np.where(key_ok, 0.0, np.finfo(np.float32).min)[:, None, None, :]
    for l in range(cfg["layers"]):
        p = f"bert.encoder.layer.{l}."
        q = x @ W[p + "attention.self.query.weight"].T + W[p + "attention.self.query.bias"]
        k = x @ W[p + "attention.self.key.weight"].T + W[p + "attention.self.key.bias"]
        v = x @ W[p + "attention.self.value.weight"].T + W[p + "attention.self.value.bias"]
        sp = lambda t: t.reshape(P, L, nh, dh).transpose(0, 2, 1, 3)
        s = sp(q) @ sp(k).transpose(0, 1, 3, 2) * (dh ** -0.5) + add_mask
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        a = e / e.sum(-1, keepdims=True)
        ctx = (a @ sp(v)).transpose(0, 2, 1, 3).reshape(P, L, H)
        o = ctx @ W[p + "attention.output.dense.weight"].T + W[p + "attention.output.dense.bias"]
        x = _ln(o + x, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], cfg["eps"])
        h = x @ W[p + "intermediate.dense.weight"].T + W[p + "intermediate.dense.bias"]
        h = 0.5 * h * (1.0 + erf(h / math.sqrt(2.0)))
        o = h @ W[p + "output.dense.weight"].T + W[p + "output.dense.bias"]
        x = _ln(o + x, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], cfg["eps"])
    return W, x
