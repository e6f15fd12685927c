#!/bin/bash
# Variant build of the library with other BM25 range / workgroup sizes: tools/bm25_variant_build.sh <BM_RANGE> <BM_THREADS>
# -> tools/bin/librag_BM_R<range>_T<threads>.so (select with RAG_HIP_LIB=, see tools/bm25_ab.sh). Needs the in-tree objects
# (python -c "import __graft_entry__ as g; g.build()") for everything but bm25.hip.
set -e
cd "$(dirname "$0")/../optimized-rag_amd/csrc"
mkdir -p ../../tools/bin
R=$1; T=$2; name=BM_R${R}_T${T}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result \
  -DBM_RANGE=$R -DBM_THREADS=$T -c bm25.hip -o /tmp/bm25_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/librag_$name.so /tmp/bm25_$name.o $(ls *.o | grep -v '^bm25.o$')
echo built tools/bin/librag_$name.so
