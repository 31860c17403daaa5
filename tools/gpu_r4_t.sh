#!/bin/bash
# (1) chained forward without its magnitude-word atomics (timing only), (2) thin first/last-layer kernels with 4096 / 1024 / 512 workgroups
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp
L=diffusion-net_amd/diffusion_net
for rep in 1 2; do for lib in libdiffnet_hip.so libdiffnet_hip_nocommit.so; do echo "=== $lib"; timeout 100 ./tools/kbench --lib $L/$lib --ops block_fwd --reps 40 2>&1 | grep -v "^#" | cut -c1-60; done; done | tee gpurun_out/t_nocommit.txt
for v in smallk4096 "" smallk512; do
  rm -rf /tmp/prof_t; DN_LIB_VARIANT=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o trace -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 < /dev/null
  f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1); echo "=== variant '$v'"; grep "smallk_rows" $f | sed 's/.*float\*)",//' | cut -c1-80
done | tee gpurun_out/t_smallk.txt
