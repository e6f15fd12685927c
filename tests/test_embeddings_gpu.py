"""GPU: the local embedding model (stands where /root/reference/memory/embeddings.py:100-115,226-246
call the OpenAI endpoint). Outside the north star's parity contract (different vectors by construction): what is pinned is the
HIP forward (the cross-encoder's kernels + mean pooling + L2 normalisation) against the float64 oracle, itself pinned to
transformers.BertModel by tests/test_oracle_bert.py. Tolerance 1e-3 per component on unit vectors, cosine to the oracle > 1 - 1e-6."""
import numpy as np
import pytest

from oracle import bert_oracle as B

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from optimized_rag_amd import RagEngine
    e = RagEngine(dim=384, device=0)
    yield e
    e.close()


def _encoder_tensors(w, cfg):
    from optimized_rag_amd.cross_encoder import flatten_state_dict
    return flatten_state_dict(w, cfg["layers"], head=False)


def test_sentence_embeddings_vs_oracle(eng):
    cfg = B.minilm_config()                                    # all-MiniLM-L6-v2 has the same shape: 6 x 384, 12 heads, FFN 1536
    w = B.seeded_weights(cfg, 321)
    eng.embed_load(cfg, _encoder_tensors(w, cfg), normalize=True)
    rng = np.random.default_rng(9)
    P, L = 200, 128
    lens = rng.integers(1, L + 1, P).astype(np.int32)
    lens[:6] = [1, 2, 15, 16, 17, L]
    ids = rng.integers(1000, cfg["vocab_size"], (P, L)).astype(np.int32)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    tt = np.zeros((P, L), dtype=np.int32)
    got = eng.embed(ids, tt, lens)
    assert got.shape == (P, 384) and np.isfinite(got).all()
    np.testing.assert_allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-5)
    sel = [0, 1, 2, 3, 4, 5, 50, 199]
    exp = B.sentence_embeddings(w, cfg, ids[sel].astype(np.int64), tt[sel].astype(np.int64), lens[sel], fast_erf=True)
    assert np.abs(got[sel] - exp).max() < 1e-3
    assert ((got[sel] * exp).sum(1) > 1 - 1e-6).all()
    # un-normalised head, device-pointer entry, and independence of the batch a text sits in
    import torch
    eng.embed_load(cfg, _encoder_tensors(w, cfg), normalize=False)
    out = torch.empty((len(sel), 384), dtype=torch.float32, device="cuda")
    eng.embed_dev(torch.from_numpy(ids[sel]).cuda(), torch.from_numpy(tt[sel]).cuda(), torch.from_numpy(lens[sel]).cuda(), out)
    torch.cuda.synchronize()
    raw = B.sentence_embeddings(w, cfg, ids[sel].astype(np.int64), tt[sel].astype(np.int64), lens[sel], normalize=False, fast_erf=True)
    assert np.abs(out.cpu().numpy() - raw).max() < 2e-3


def test_local_embedding_service_surface_and_index_round_trip(eng, tmp_path):
    """LocalEmbeddingService (the reference EmbeddingService's methods) from a local checkpoint directory feeds a 384-d index:
    every text finds itself first, the cache answers repeats, batch == single."""
    import json
    from safetensors.numpy import save_file
    from optimized_rag_amd import RagEngine
    from optimized_rag_amd.embeddings import LocalEmbeddingService
    cfg = dict(vocab_size=60, hidden=128, layers=2, heads=4, ffn=256, max_pos=64, type_vocab=2, eps=1e-12)
    w = B.seeded_weights(cfg, 5)
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(55)]
    d = tmp_path / "encoder"
    d.mkdir()
    (d / "vocab.txt").write_text("\n".join(words) + "\n")
    (d / "config.json").write_text(json.dumps(dict(vocab_size=60, hidden_size=128, num_hidden_layers=2, num_attention_heads=4,
                                                   intermediate_size=256, max_position_embeddings=64, type_vocab_size=2,
                                                   hidden_act="gelu", layer_norm_eps=1e-12)))
    save_file({k[len("bert."):]: v for k, v in w.items() if k.startswith("bert.") and "pooler" not in k}, str(d / "model.safetensors"))
    e128 = RagEngine(dim=128, device=0)
    try:
        svc = LocalEmbeddingService.from_dir(str(d), engine=e128)
        assert svc.get_embedding_dimension() == 128
        rng = np.random.default_rng(1)
        texts = [" ".join(rng.choice(words[5:], int(rng.integers(3, 20)))) for _ in range(300)]
        vecs = np.asarray(svc.generate_embeddings_batch(texts), dtype=np.float32)
        one = np.asarray(svc.generate_embedding(texts[7], use_cache=False), dtype=np.float32)
        np.testing.assert_array_equal(one, vecs[7])                                  # a text's vector does not depend on its batch
        assert svc.generate_embedding(texts[7]) == [float(x) for x in vecs[7]] and svc.get_cache_stats()["hits"] >= 1
        with pytest.raises(ValueError):
            svc.generate_embedding("   ")
        # ADVICE r3, the reference's surface (memory/embeddings.py:164-168, :50, :270-290): an empty item of a batch gets [] in its
        # slot and counts as neither hit nor miss; the cache is an LRU of max_size entries
        before = svc.get_cache_stats()
        mixed = svc.generate_embeddings_batch([texts[0], "", "   ", texts[1]])
        after = svc.get_cache_stats()
        assert mixed[1] == [] and mixed[2] == [] and mixed[0] == [float(x) for x in vecs[0]] and mixed[3] == [float(x) for x in vecs[1]]
        assert after["hits"] - before["hits"] == 2 and after["misses"] == before["misses"]
        assert after["max_size"] == 1000 and after["current_size"] <= 1000 and "hit_rate" in after and after["cache_full"] is False
        small = LocalEmbeddingService.from_dir(str(d), engine=e128, cache_size=4)
        small.generate_embeddings_batch(texts[:10])
        st = small.get_cache_stats()
        assert st["current_size"] == 4 and st["max_size"] == 4 and st["cache_full"] is True
        assert small.generate_embedding(texts[9]) == [float(x) for x in vecs[9]] and small.get_cache_stats()["hits"] == 1   # most recent kept
        ids, tt, lens = svc.tokenize(texts[:4])
        exp = B.sentence_embeddings(w, cfg, ids.astype(np.int64), tt.astype(np.int64), lens)
        assert np.abs(vecs[:4] - exp).max() < 1e-3
        e128.index_load(vecs)
        got, _, sc = e128.dense_topk(vecs[:50], 1)
        uniq = {t: i for i, t in reversed(list(enumerate(texts)))}                   # duplicates: the first occurrence wins ties
        assert got[:, 0].tolist() == [uniq[t] for t in texts[:50]] and (sc[:, 0] > 1 - 1e-6).all()
    finally:
        e128.close()
