#!/bin/bash
# size sweep of the chained kernels (forward / backward, on / off) for the enabling thresholds; cfg2 / cfg5 / cfg3 bench lines
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for cfg in "--meshes 1 --verts 3000" "--meshes 1 --verts 7000" "--meshes 1 --verts 14000" "--meshes 2 --verts 10000" "--meshes 4 --verts 10000" "--meshes 8 --verts 10000"; do
  for ch in 1 0; do
    echo "=== $cfg chain=$ch"; DN_CHAIN=$ch DN_CHAIN_MIN_ROWS=0 timeout 120 ./tools/kbench $cfg --ops block_inf,block_fwd,block_bwd --reps 50 2>&1 | grep -v "^#" | cut -c1-40
  done
done 2>&1 | tee gpurun_out/g_sweep.txt
for ch in 1 0; do
  DN_CHAIN=$ch DN_CHAIN_MIN_ROWS=0 timeout 300 python bench.py --config cfg2 --steps 40 2>> gpurun_out/g_bench.err | python tools/bench_brief.py
  DN_CHAIN=$ch DN_CHAIN_MIN_ROWS=0 timeout 300 python bench.py --config cfg2 --graph --steps 40 2>> gpurun_out/g_bench.err | python tools/bench_brief.py
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg5 --steps 10 2>> gpurun_out/g_bench.err | python tools/bench_brief.py
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg3 --graph --steps 20 2>> gpurun_out/g_bench.err | python tools/bench_brief.py
done
