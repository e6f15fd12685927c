"""Host side of the cross-encoder (K7): local checkpoint loading, WordPiece tokenisation of (query, doc) pairs
and batching into rag_ce_score_host. Stands where `sentence_transformers.CrossEncoder` stands in the reference
(/root/reference/rag/reranker.py:312-313,355): `predict(pairs)` returns RAW logits (the ms-marco checkpoints use an
identity activation, which is why the reference applies its own sigmoid at :359).

Weights are read with `safetensors` (numpy) and handed to the HIP engine as float32 arrays; PyTorch is not needed.
"""
import json
import os

import numpy as np

from .engine import get_engine

LAYER_KEYS = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
              "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
              "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
              "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
              "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]


def flatten_state_dict(sd, n_layers, head=True, prefix="bert."):
    """HF BertForSequenceClassification state dict -> the tensor order rag_ce_load_host expects. head=False: a plain BertModel
    encoder for rag_embed_load_host (no pooler / classifier; sentence-transformers checkpoints store it without the `bert.`
    prefix: prefix="")."""
    names = [prefix + "embeddings.word_embeddings.weight", prefix + "embeddings.position_embeddings.weight",
             prefix + "embeddings.token_type_embeddings.weight", prefix + "embeddings.LayerNorm.weight",
             prefix + "embeddings.LayerNorm.bias"]
    for l in range(n_layers):
        names += [f"{prefix}encoder.layer.{l}.{k}" for k in LAYER_KEYS]
    if head:
        names += [prefix + "pooler.dense.weight", prefix + "pooler.dense.bias", "classifier.weight", "classifier.bias"]
    missing = [n for n in names if n not in sd]
    if missing:
        raise KeyError(f"checkpoint lacks {missing[:3]}{'...' if len(missing) > 3 else ''}")
    return [np.ascontiguousarray(np.asarray(sd[n], dtype=np.float32)) for n in names]


def config_from_hf(cfg):
    """config.json of the checkpoint -> engine config (shape comes from the checkpoint, nothing is hard-coded)."""
    if cfg.get("hidden_act", "gelu") != "gelu":
        raise ValueError("only exact-erf GELU checkpoints are supported")
    return dict(vocab_size=cfg["vocab_size"], hidden=cfg["hidden_size"], layers=cfg["num_hidden_layers"],
                heads=cfg["num_attention_heads"], ffn=cfg["intermediate_size"], max_pos=cfg["max_position_embeddings"],
                type_vocab=cfg.get("type_vocab_size", 2), eps=cfg.get("layer_norm_eps", 1e-12))


MINILM_L6_CONFIG = dict(vocab_size=30522, hidden=384, layers=6, heads=12, ffn=1536, max_pos=512, type_vocab=2, eps=1e-12)
"""Shape of cross-encoder/ms-marco-MiniLM-L-6-v2 (the checkpoint the reference names at config.py:49), for benchmarks
that run without the downloaded weights; a real deployment takes the shape from the checkpoint's config.json."""


def random_init_tensors(cfg, seed=0):
    """Random-init weights of the architecture in rag_ce_load_host's tensor order (benchmarks have no checkpoint: there
    is no network). Scales are those of a trained BERT (0.02-0.1), LayerNorm gains around 1."""
    rng = np.random.default_rng(seed)
    H, F = cfg["hidden"], cfg["ffn"]

    def mat(*shape, s=0.05):
        return (rng.standard_normal(shape) * s).astype(np.float32)

    def gain():
        return (1.0 + mat(H, s=0.1)).astype(np.float32)

    out = [mat(cfg["vocab_size"], H, s=0.1), mat(cfg["max_pos"], H, s=0.1), mat(cfg.get("type_vocab", 2), H, s=0.1), gain(), mat(H, s=0.1)]
    for _ in range(cfg["layers"]):
        out += [mat(H, H, s=0.08), mat(H), mat(H, H, s=0.08), mat(H), mat(H, H, s=0.08), mat(H),      # q, k, v
                mat(H, H), mat(H), gain(), mat(H, s=0.1),                                              # attention output + LN
                mat(F, H), mat(F), mat(H, F), mat(H), gain(), mat(H, s=0.1)]                           # FFN + LN
    out += [mat(H, H), mat(H), mat(1, H, s=0.5), mat(1, s=0.5)]                                        # pooler, classifier
    return out


class LocalCrossEncoder:
    def __init__(self, cfg, tensors, tokenizer, max_length=512, engine=None, batch_pairs=4096):
        self.cfg = cfg
        self.engine = engine or get_engine()
        self.tokenizer = tokenizer
        self.max_length = min(int(max_length), cfg["max_pos"], 512)
        self.batch_pairs = batch_pairs
        self.engine.ce_load(cfg, tensors)

    @classmethod
    def from_dir(cls, path, max_length=512, engine=None):
        from safetensors.numpy import load_file
        from tokenizers import BertWordPieceTokenizer
        with open(os.path.join(path, "config.json")) as f:
            cfg = config_from_hf(json.load(f))
        sd = load_file(os.path.join(path, "model.safetensors"))
        lower = True
        tk_cfg = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(tk_cfg):
            with open(tk_cfg) as f:
                lower = bool(json.load(f).get("do_lower_case", True))
        tok = BertWordPieceTokenizer(os.path.join(path, "vocab.txt"), lowercase=lower)
        return cls(cfg, flatten_state_dict(sd, cfg["layers"]), tok, max_length=max_length, engine=engine)

    def tokenize_pairs(self, pairs):
        """[CLS] q [SEP] d [SEP], token_type 0/1, truncation 'longest_first' to max_length, padded to the longest."""
        self.tokenizer.enable_truncation(max_length=self.max_length, strategy="longest_first")
        self.tokenizer.no_padding()
        enc = self.tokenizer.encode_batch([(str(q), str(d)) for q, d in pairs])
        L = max(len(e.ids) for e in enc)
        ids = np.zeros((len(enc), L), dtype=np.int32)
        tt = np.zeros((len(enc), L), dtype=np.int32)
        lens = np.zeros((len(enc),), dtype=np.int32)
        for i, e in enumerate(enc):
            n = len(e.ids)
            ids[i, :n] = e.ids
            tt[i, :n] = e.type_ids
            lens[i] = n
        return ids, tt, lens

    def predict(self, pairs):
        out = np.empty((len(pairs),), dtype=np.float32)
        for b in range(0, len(pairs), self.batch_pairs):
            ids, tt, lens = self.tokenize_pairs(pairs[b:b + self.batch_pairs])
            out[b:b + len(lens)] = self.engine.ce_score(ids, tt, lens)
        return out
