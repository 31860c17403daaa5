#!/bin/bash
# TN products (projections, weight gradients) with nontemporal operand loads (variant tnstream) vs plain
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
for rep in 1 2; do
for lib in libdiffnet_hip.so libdiffnet_hip_tnstream.so; do echo "=== $lib"; timeout 200 ./tools/kbench --lib $L/$lib --ops diffusion,block_fwd,block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70; done
done | tee gpurun_out/q_kbench.txt
for v in "" tnstream; do DN_LIB_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/q_bench.err < /dev/null | python tools/bench_brief.py; done | tee gpurun_out/q_bench.txt
