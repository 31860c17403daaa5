#!/bin/bash
# small batches: workgroup width chosen by batch size (NW = 1 / 2 / 4), prep kernel doing the start-of-call bookkeeping
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for cfg in "--meshes 1 --verts 7000" "--meshes 2 --verts 10000" "--meshes 16 --verts 10000"; do
  for nw in 0 1 2 4; do
    echo "=== kbench $cfg DN_CHAIN_NW=$nw"; DN_CHAIN_NW=$nw timeout 120 ./tools/kbench $cfg --ops block_inf,block_fwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60
  done
  echo "=== kbench $cfg DN_CHAIN=0"; DN_CHAIN=0 timeout 120 ./tools/kbench $cfg --ops block_inf,block_fwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60
done 2>&1 | tee gpurun_out/d_kbench.txt
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg2 --steps 40 2>> gpurun_out/d_bench.err | python tools/bench_brief.py
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg2 --graph --steps 40 2>> gpurun_out/d_bench.err | python tools/bench_brief.py
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or ragged or headline or inkernel or autograph" 2>&1 | tail -3
