#!/bin/bash
# chained backward kernel: kbench A/B, parity subset, bench A/B
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for ch in 1 0; do
  echo "=== kbench DN_CHAIN=$ch (16 x 10k)"; DN_CHAIN=$ch timeout 120 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 30 2>&1 | grep -v "^#" | cut -c1-200
done 2>&1 | tee gpurun_out/f_kbench.txt
DN_CHAIN_MIN_ROWS=0 timeout 120 ./tools/kbench --meshes 1 --verts 7000 --ops block_fwd,block_bwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60 | tee -a gpurun_out/f_kbench.txt
DN_CHAIN=0 timeout 120 ./tools/kbench --meshes 1 --verts 7000 --ops block_fwd,block_bwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60 | tee -a gpurun_out/f_kbench.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or golden or ragged or headline or inkernel or wide_head or dropout_masks or determinism" 2>&1 | tail -3
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/f_bench_chain$ch.json 2> gpurun_out/f_bench.err < /dev/null
  python tools/bench_brief.py < gpurun_out/f_bench_chain$ch.json
done
