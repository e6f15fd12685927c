"""Process-wide default engine (one GPU per process) used by the mirror classes when none is passed."""
import os

from ._lib import RagEngine

_DEFAULT = {}


def get_engine(dim=1536, device=None):
    """Lazily create the engine for this process's GPU. Raises RagError when the HIP library or the GPU is
    missing — there is no CPU path to fall back to."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    key = (int(dim), int(device))
    if key not in _DEFAULT:
        _DEFAULT[key] = RagEngine(dim=dim, device=device)
    return _DEFAULT[key]


def as_matrix(vectors):
    """List[List[float]] (possibly ragged) -> float64 [n, d] zero-padded to the longest row.
    Zero padding reproduces the reference's `zip()` truncation exactly: the dot product runs over the shorter
    length while each norm uses the full vector (rag/retrieval.py:364-366). float64, as the reference's Python floats: the
    cosine kernel takes them unrounded (rag_pairwise_cosine_f64_host); the float32-only entries (MMR, chunk chain) cast."""
    import numpy as np
    rows = [np.asarray(v if v is not None else [], dtype=np.float64).ravel() for v in vectors]
    d = max([r.shape[0] for r in rows] + [1])
    d = (d + 3) // 4 * 4
    out = np.zeros((len(rows), d), dtype=np.float64)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out
