#!/bin/bash
# chained kernel v2 (deep LDS-DMA ring): kbench NW=4/8, phase trace
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for nw in 4 8; do
  echo "=== kbench DN_CHAIN_NW=$nw (16 x 10k)"
  DN_CHAIN_NW=$nw timeout 120 ./tools/kbench --ops block_inf,block_fwd --reps 30 2>&1 | grep -v "^#" | cut -c1-200
done 2>&1 | tee gpurun_out/b_kbench.txt
echo "=== kbench 1 x 7000"; for nw in 4 8; do DN_CHAIN_NW=$nw timeout 120 ./tools/kbench --meshes 1 --verts 7000 --ops block_inf,block_fwd --reps 50 2>&1 | grep -v "^#" | cut -c1-80; done 2>&1 | tee -a gpurun_out/b_kbench.txt
for nw in 4 8; do
echo "=== trace NW=$nw block_inf"; DN_CHAIN_NW=$nw timeout 120 ./tools/kbench --lib diffusion-net_amd/diffusion_net/libdiffnet_hip_chtrace.so --ops block_inf --reps 3 --trace 2>&1 | grep -v "^#" | tee -a gpurun_out/b_trace.txt
echo "=== trace NW=$nw block_fwd"; DN_CHAIN_NW=$nw timeout 120 ./tools/kbench --lib diffusion-net_amd/diffusion_net/libdiffnet_hip_chtrace.so --ops block_fwd --reps 3 --trace 2>&1 | grep -v "^#" | tee -a gpurun_out/b_trace.txt
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or ragged or headline or inkernel" 2>&1 | tail -3
