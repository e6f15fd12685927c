# Diagnostic: builds tools/ce_mx_probe.hip (optionally in its timing-experiment variants, DIAGS="0 1 2 4 8") and runs it (GPU box)
set -e
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out tools/bin
for d in ${DIAGS:-0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMX_DIAG=$d $EXTRA tools/ce_mx_probe.hip -o tools/bin/ce_mx_probe_d$d 2>/dev/null
  for ds in ${DESYNC:-0}; do
  echo "== MX_DIAG=$d (1 no DMA, 2 no MFMA, 4 no weight DMA, 8 no token DMA) desync $ds $EXTRA"
  timeout -k 10 120 tools/bin/ce_mx_probe_d$d 1048576 1152 384 0 $ds | grep -E "main loop|epilogue:"
  timeout -k 10 120 tools/bin/ce_mx_probe_d$d 1048576 384 1536 0 $ds | grep -E "main loop|epilogue:"
  done
done
