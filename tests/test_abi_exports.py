"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/rag_hip.h declares.
No compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as G


@pytest.fixture(scope="module")
def lib():
    G.build()
    import optimized_rag_amd
    return optimized_rag_amd.load_library()


def test_header_symbols_all_exported(lib):
    from optimized_rag_amd import _lib
    hdr = open(os.path.join(G.ROOT, "include", "rag_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rag_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    assert declared == _lib.exported_symbols(), "ctypes signature table and header disagree"
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by librag_hip.so"


def test_version_and_null_handle(lib):
    assert lib.rag_version() >= 100
    assert lib.rag_destroy(None) != 0
    assert lib.rag_last_error(None) == b"null handle"


def test_no_cpu_fallback_without_gpu(lib):
    """Without a GPU, creating an engine must fail loudly rather than fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from optimized_rag_amd import RagEngine, RagError
    with pytest.raises(RagError):
        RagEngine(dim=1536, device=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(G.ROOT, "optimized-rag_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the float64 oracle", "").replace("oracle's", ""), f


def test_bench_reads_the_committed_traffic_profile():
    """bench.py fills roofline.traffic from profiles/r02_g_dense_pmc.json: the file must carry the key it reads (a reshaped
    file once took the default bench line down; bench.py now also tolerates it)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r02_g_dense_pmc.json")) as f:
        total = json.load(f)["kernels"]["dense_emit_kernel<false>"]["hbm_traffic_bytes_per_launch"]["total"]
    assert 1.0e9 < total < 3.0e9            # ~1.5 GB per launch against 1.025 GB algorithmic
