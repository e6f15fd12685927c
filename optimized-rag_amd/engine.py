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
    """List[List[float]] (possibly ragged) -> float32 [n, d] zero-padded to the longest row.
    Zero padding reproduces the reference's `zip()` truncation exactly: the dot product runs over the shorter
    length while each norm uses the full vector (rag/retrieval.py:364-366)."""
    import numpy as np
    rows = [np.asarray(v if v is not None else [], dtype=np.float32).ravel() for v in vectors]
    d = max([r.shape[0] for r in rows] + [1])
    d = (d + 3) // 4 * 4
    out = np.zeros((len(rows), d), dtype=np.float32)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out
