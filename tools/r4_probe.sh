# Diagnostic: builds tools/ce_mx_probe.hip in its timing-experiment variants and runs them (GPU box)
set -e
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out tools/bin
for d in 0 1 2 4 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMX_DIAG=$d tools/ce_mx_probe.hip -o tools/bin/ce_mx_probe_d$d 2>/dev/null
  echo "== MX_DIAG=$d (1 no DMA, 2 no MFMA, 4 no weight DMA, 8 no token DMA)"
  timeout -k 10 120 tools/bin/ce_mx_probe_d$d 1048576 1152 384 0 | grep -E "main loop|image epi"
  timeout -k 10 120 tools/bin/ce_mx_probe_d$d 1048576 384 1536 0 | grep -E "main loop|image epi"
done
