#!/bin/bash
# Diagnostic builds of librag_hip.so with parts of the cross-encoder GEMM loop removed (timing experiments; results are
# wrong by construction). Output: tools/bin/librag_<variant>.so, selected at run time with RAG_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../optimized-rag_amd/csrc"
mkdir -p ../../tools/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result"
OTHERS=$(ls *.o | grep -v cross_encoder.o)
VARIANTS=${CE_PROBE_VARIANTS:-'NO_MFMA NO_DMA NO_EPI NO_MFMA+NO_EPI NO_DMA+NO_EPI G12_NO_STORE G12_NO_GELU G12_NO_STORE+G12_NO_GELU'}
for vv in $VARIANTS; do
  v=$(echo "$vv" | sed 's/+/ -DCE_PROBE_/g; s/CE_PROBE_G12_/G12_PROBE_/g')
  case "$v" in G12_*) v=$(echo "$v" | sed 's/^G12_/G12_PROBE_/'); D="-D$v";; *) D="-DCE_PROBE_$v";; esac
  name=$(echo "$vv" | tr '+' '_')
  /opt/rocm/bin/hipcc $FLAGS $D -c cross_encoder.hip -o /tmp/ce_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/librag_$name.so /tmp/ce_$name.o $OTHERS
  echo built $name
done
