"""Local embedding service on the GPU: the class surface of the reference's `EmbeddingService`
(/root/reference/memory/embeddings.py: generate_embedding :64-98, generate_embeddings_batch :154-224, get_embedding_dimension
:312-332, cache statistics :248-310) with the OpenAI HTTP calls (:100-115, :226-246) replaced by a BERT sentence encoder run by
the HIP engine (rag_embed_load_host / rag_embed_host: the cross-encoder's kernels behind a mean-pooling + L2-normalise head,
i.e. a sentence-transformers `Transformer -> Pooling(mean) -> Normalize` checkpoint such as all-MiniLM-L6-v2).

NOT a parity replacement: a local encoder returns different vectors (and a different dimension: 384 for the MiniLM shape) than
text-embedding-3-small. An index has to be built and queried with the same service; what the tests pin is this forward against
`transformers.BertModel`. There is no CPU fallback.
"""
import json
import os
import threading
from collections import OrderedDict

import numpy as np

from .cross_encoder import config_from_hf, flatten_state_dict
from .engine import get_engine


class LocalEmbeddingService:
    def __init__(self, cfg, tensors, tokenizer, max_length=256, engine=None, batch_size=2048, normalize=True, model="local-bert-mean-pool",
                 cache_size=1000):
        self.cfg = cfg
        self.model = model
        self.dimensions = int(cfg["hidden"])
        self.engine = engine or get_engine(dim=self.dimensions)
        self.tokenizer = tokenizer
        self.max_length = min(int(max_length), cfg["max_pos"], 512)
        self.batch_size = int(batch_size)
        self.engine.embed_load(cfg, tensors, normalize=normalize)
        # LRU cache of `cache_size` texts (the reference: cachetools.LRUCache(maxsize=EMBEDDING_CACHE_SIZE), default 1000 -
        # memory/embeddings.py:50, config.py:77): ingesting a large corpus must not keep every text in host memory
        self._cache, self._cache_lock = OrderedDict(), threading.Lock()
        self._cache_max = max(1, int(cache_size))
        self._cache_hits = self._cache_misses = 0

    def _cache_get(self, text):                      # caller holds the lock
        hit = self._cache.get(text)
        if hit is not None:
            self._cache.move_to_end(text)
        return hit

    def _cache_put(self, text, emb):                 # caller holds the lock
        self._cache[text] = emb
        self._cache.move_to_end(text)
        while len(self._cache) > self._cache_max:
            self._cache.popitem(last=False)

    @classmethod
    def from_dir(cls, path, max_length=256, engine=None, **kw):
        """A local sentence-transformers / HF BertModel directory: config.json, model.safetensors, vocab.txt."""
        from safetensors.numpy import load_file
        from tokenizers import BertWordPieceTokenizer
        with open(os.path.join(path, "config.json")) as f:
            cfg = config_from_hf(json.load(f))
        sd = load_file(os.path.join(path, "model.safetensors"))
        prefix = "bert." if any(k.startswith("bert.") for k in sd) else ""
        lower = True
        tk_cfg = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(tk_cfg):
            with open(tk_cfg) as f:
                lower = bool(json.load(f).get("do_lower_case", True))
        tok = BertWordPieceTokenizer(os.path.join(path, "vocab.txt"), lowercase=lower)
        return cls(cfg, flatten_state_dict(sd, cfg["layers"], head=False, prefix=prefix), tok, max_length=max_length, engine=engine,
                   model=os.path.basename(os.path.normpath(path)), **kw)

    # ---- tokenisation: [CLS] text [SEP], truncated to max_length, padded to the longest of the batch -----------------
    def tokenize(self, texts):
        self.tokenizer.enable_truncation(max_length=self.max_length)
        self.tokenizer.no_padding()
        enc = self.tokenizer.encode_batch([str(t) for t in texts])
        L = max(len(e.ids) for e in enc)
        ids = np.zeros((len(enc), L), dtype=np.int32)
        lens = np.zeros((len(enc),), dtype=np.int32)
        for i, e in enumerate(enc):
            ids[i, :len(e.ids)] = e.ids
            lens[i] = len(e.ids)
        return ids, np.zeros_like(ids), lens

    def _embed_uncached(self, texts):
        out = np.empty((len(texts), self.dimensions), dtype=np.float32)
        for b in range(0, len(texts), self.batch_size):
            ids, tt, lens = self.tokenize(texts[b:b + self.batch_size])
            out[b:b + len(lens)] = self.engine.embed(ids, tt, lens)
        return out

    # ---- the reference's surface ------------------------------------------------------------------------------------
    def generate_embedding(self, text, use_cache=True):
        if not text or not text.strip():
            raise ValueError("Text cannot be empty")                              # memory/embeddings.py:75-76
        if use_cache:
            with self._cache_lock:
                hit = self._cache_get(text)
                if hit is not None:
                    self._cache_hits += 1
                    return list(hit)
                self._cache_misses += 1
        emb = [float(x) for x in self._embed_uncached([text])[0]]
        if use_cache:
            with self._cache_lock:
                self._cache_put(text, tuple(emb))
        return emb

    def generate_embeddings_batch(self, texts, use_cache=True):
        """List[str] -> List[List[float]] in input order; cached texts are not recomputed; an empty or whitespace-only text gets [] in
        its slot and touches neither the cache nor its counters (memory/embeddings.py:154-224, :164-168)."""
        if not texts:
            return []
        out, todo = [[] for _ in texts], []
        for i, t in enumerate(texts):
            if not t or not t.strip():
                continue
            hit = None
            if use_cache:
                with self._cache_lock:
                    hit = self._cache_get(t)
                    if hit is not None:
                        self._cache_hits += 1
                    else:
                        self._cache_misses += 1
            if hit is not None:
                out[i] = list(hit)
            else:
                todo.append(i)
        if todo:
            vecs = self._embed_uncached([texts[i] for i in todo])
            for i, v in zip(todo, vecs):
                out[i] = [float(x) for x in v]
                if use_cache:
                    with self._cache_lock:
                        self._cache_put(texts[i], tuple(out[i]))
        return out

    def get_embedding_dimension(self):
        return self.dimensions

    def get_cache_stats(self):
        with self._cache_lock:
            total = self._cache_hits + self._cache_misses
            rate = self._cache_hits / total if total else 0.0
            return {"hits": self._cache_hits, "misses": self._cache_misses, "hit_rate": rate, "hit_rate_percent": f"{rate * 100:.1f}%",
                    "current_size": len(self._cache), "max_size": self._cache_max, "cache_full": len(self._cache) >= self._cache_max}

    def clear_cache(self):
        with self._cache_lock:
            self._cache.clear()
            self._cache_hits = self._cache_misses = 0
