#!/bin/bash
# round 4, first measurement of the chained row kernel: kbench block timings with the chain on / off, GPU parity tier, short bench
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for ch in 1 0; do
  echo "=== kbench DN_CHAIN=$ch (16 x 10k)"
  DN_CHAIN=$ch timeout 120 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 30 2>&1 | grep -v "^#" | cut -c1-200
done 2>&1 | tee gpurun_out/a_kbench.txt
echo "=== kbench 1 x 7000"; for ch in 1 0; do DN_CHAIN=$ch timeout 120 ./tools/kbench --meshes 1 --verts 7000 --ops block_inf,block_fwd --reps 50 2>&1 | grep -v "^#" | cut -c1-80; done 2>&1 | tee -a gpurun_out/a_kbench.txt
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/a_gpu_tests.log 2>&1 < /dev/null; tail -5 gpurun_out/a_gpu_tests.log
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/a_bench_chain$ch.json 2> gpurun_out/a_bench.err < /dev/null
  python tools/bench_brief.py < gpurun_out/a_bench_chain$ch.json
done
