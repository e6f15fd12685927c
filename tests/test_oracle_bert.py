"""Pin oracle/bert_oracle.py against the transformers BertForSequenceClassification shipped in the image
(real ms-marco-MiniLM weights are unavailable offline -> seeded weights) and the committed fixture."""
import os

import numpy as np
import pytest

from oracle import bert_oracle as B


def test_bert_oracle_matches_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "bert_minilm.npz"))
    cfg = B.minilm_config()
    w = B.seeded_weights(cfg, int(g["seed"]))
    sel = slice(0, 4)
    lg = B.forward_logits(w, cfg, g["input_ids"][sel].astype(np.int64), g["token_type_ids"][sel].astype(np.int64), g["lens"][sel])
    np.testing.assert_allclose(lg, g["logits"][sel], atol=2e-4)     # fixture is torch fp32


def test_bert_oracle_matches_transformers_small():
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    cfg = dict(vocab_size=200, hidden=64, layers=2, heads=4, ffn=128, max_pos=40, type_vocab=2, eps=1e-12)
    w = B.seeded_weights(cfg, 7)
    hf = tr.BertForSequenceClassification(tr.BertConfig(
        vocab_size=200, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
        max_position_embeddings=40, type_vocab_size=2, hidden_act="gelu", layer_norm_eps=1e-12, num_labels=1)).eval().double()
    sd = hf.state_dict()
    for k, v in w.items():
        sd[k].copy_(torch.from_numpy(v).double())
    rng = np.random.default_rng(3)
    P, L = 5, 24
    lens = np.array([24, 5, 17, 2, 11])
    ids = rng.integers(1, 200, (P, L))
    tt = (np.arange(L)[None] >= 4).astype(np.int64) * np.ones((P, 1), dtype=np.int64)
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        ref = hf(input_ids=torch.from_numpy(ids), token_type_ids=torch.from_numpy(tt),
                 attention_mask=torch.from_numpy(mask)).logits[:, 0].numpy()
    mine = B.forward_logits(w, cfg, ids, tt, lens)
    np.testing.assert_allclose(mine, ref, atol=1e-10)
    # the vectorised-erf path the full-size GPU tests use is the same function
    np.testing.assert_allclose(B.forward_logits(w, cfg, ids, tt, lens, fast_erf=True), mine, atol=1e-12)


def test_sentence_embedding_oracle_matches_transformers_bert_model():
    """the embedding model's oracle = transformers.BertModel (double) + mean pooling over the real tokens +
    torch.nn.functional.normalize - what a sentence-transformers Transformer -> Pooling(mean) -> Normalize checkpoint computes."""
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    cfg = dict(vocab_size=300, hidden=128, layers=2, heads=4, ffn=256, max_pos=40, type_vocab=2, eps=1e-12)
    w = B.seeded_weights(cfg, 11)
    hf = tr.BertModel(tr.BertConfig(
        vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256,
        max_position_embeddings=40, type_vocab_size=2, hidden_act="gelu", layer_norm_eps=1e-12), add_pooling_layer=False).eval().double()
    sd = hf.state_dict()
    for k in sd:
        sd[k].copy_(torch.from_numpy(w["bert." + k]).double())
    rng = np.random.default_rng(5)
    P, L = 6, 30
    lens = np.array([30, 1, 17, 2, 11, 29])
    ids = rng.integers(1, 300, (P, L))
    tt = np.zeros((P, L), dtype=np.int64)
    mask = (np.arange(L)[None] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        hs = hf(input_ids=torch.from_numpy(ids), token_type_ids=torch.from_numpy(tt), attention_mask=torch.from_numpy(mask)).last_hidden_state
        m = torch.from_numpy(mask).double()[:, :, None]
        ref = torch.nn.functional.normalize((hs * m).sum(1) / m.sum(1).clamp(min=1e-9), p=2, dim=1).numpy()
    mine = B.sentence_embeddings(w, cfg, ids, tt, lens)
    np.testing.assert_allclose(mine, ref, atol=1e-10)
    raw = B.sentence_embeddings(w, cfg, ids, tt, lens, normalize=False)
    np.testing.assert_allclose(raw / np.linalg.norm(raw, axis=1, keepdims=True), mine, atol=1e-12)
