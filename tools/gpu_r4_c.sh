#!/bin/bash
# chained kernel in the training step: bench A/B (chain on / off), GPU tier with the chain on
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/c_bench_chain$ch.json 2> gpurun_out/c_bench.err < /dev/null
  python tools/bench_brief.py < gpurun_out/c_bench_chain$ch.json
done
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg2 --steps 40 2>> gpurun_out/c_bench.err | python tools/bench_brief.py
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg5 --steps 10 2>> gpurun_out/c_bench.err | python tools/bench_brief.py
done
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/c_gpu_tests.log 2>&1 < /dev/null; tail -3 gpurun_out/c_gpu_tests.log
