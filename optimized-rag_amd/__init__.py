"""MI355X-native hybrid retrieval + rerank engine behind the reference's Python class surface.

Host side mirrors /root/reference/rag/{retrieval,reranker,consistency_checker,context_compressor}.py and
rag/nodes/helpers.py::apply_mmr; all arithmetic runs in hand-written HIP (csrc/) through the C-ABI of
librag_hip.so (include/rag_hip.h). There is no CPU fallback: without the built library every operation raises.
"""
from ._lib import RagEngine, RagError, lib_path, load_library  # noqa: F401

__all__ = ["RagEngine", "RagError", "lib_path", "load_library"]
