#!/bin/bash
# spmm with streaming outputs vs plain (variant spplain)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
for rep in 1 2; do
for lib in libdiffnet_hip_spplain.so libdiffnet_hip.so; do echo "=== $lib"; timeout 200 ./tools/kbench --lib $L/$lib --ops spmm,block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70; done
done | tee gpurun_out/p_kbench.txt
for v in spplain ""; do DN_LIB_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/p_bench.err < /dev/null | python tools/bench_brief.py; done | tee gpurun_out/p_bench.txt
