#!/bin/bash
# kbench A/B of library variants (same sources, other build flags).  VARIANTS="_d2 _ko1 ..." (library suffixes), OPS, ROUNDS
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
for round in $(seq 1 ${ROUNDS:-1}); do for v in "" ${VARIANTS:-_d2}; do
  echo "=== variant '${v}' round $round"
  timeout 120 ./tools/kbench --lib $L/libdiffnet_hip${v}.so --ops ${OPS:-from_basis,linear,linear_relu,block_fwd} --reps ${REPS:-20} 2>&1 | grep -v "^#" | cut -c1-60
done; done | tee gpurun_out/e_ab.txt
