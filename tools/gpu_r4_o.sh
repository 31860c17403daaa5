#!/bin/bash
# chained kernels with nontemporal activation stores / single-use loads vs plain (variant nostream)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
for rep in 1 2; do
for lib in libdiffnet_hip_nostream.so libdiffnet_hip.so; do echo "=== $lib"; timeout 200 ./tools/kbench --lib $L/$lib --ops block_inf,block_fwd,block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70; done
done | tee gpurun_out/o_kbench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or golden or headline" 2>&1 | tail -3 | tee gpurun_out/o_tests.txt
for v in nostream ""; do DN_LIB_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/o_bench.err < /dev/null | python tools/bench_brief.py; done | tee gpurun_out/o_bench.txt
