#!/bin/bash
# Diagnostic builds of librag_hip.so with parts of the cross-encoder GEMM loop removed (timing experiments; results are
# wrong by construction). Output: tools/bin/librag_<variant>.so, selected at run time with RAG_HIP_LIB=<path>.
set -e
cd "$(dirname "$0")/../optimized-rag_amd/csrc"
mkdir -p ../../tools/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result"
OTHERS=$(ls *.o | grep -v cross_encoder.o)
# the split-term ablation build (RAG_CE_TERMS, tools/ce_ablation.py): the product library ships only the full-term kernels
/opt/rocm/bin/hipcc $FLAGS -DRAG_CE_ABLATION -c cross_encoder.hip -o /tmp/ce_ablation.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/librag_ablation.so /tmp/ce_ablation.o $OTHERS
echo built ablation
[ "$1" = "ablation" ] && exit 0
# fused-FFN probe: phase A's token slices always from tile 0 (L2-hot): is the kernel bound by re-fetching its token tile?
/opt/rocm/bin/hipcc $FLAGS -DCE_PROBE_FFN_SHARED_X -c cross_encoder.hip -o /tmp/ce_ffn_shared_x.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/librag_ffn_shared_x.so /tmp/ce_ffn_shared_x.o $OTHERS
echo built ffn_shared_x
[ "$1" = "ffn" ] && exit 0
for v in NO_MFMA NO_DMA NO_EPI "NO_MFMA -DCE_PROBE_NO_EPI" "NO_DMA -DCE_PROBE_NO_EPI"; do
  name=$(echo "$v" | tr -d ' ' | sed 's/-DCE_PROBE_/_/g')
  /opt/rocm/bin/hipcc $FLAGS -DCE_PROBE_$v -c cross_encoder.hip -o /tmp/ce_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/librag_$name.so /tmp/ce_$name.o $OTHERS
  echo built $name
done
