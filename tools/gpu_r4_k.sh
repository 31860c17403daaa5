#!/bin/bash
# chained forward with the row-contiguous gather: kernel times, chain parity tests, one bench line
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 200 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70 | tee gpurun_out/k_kbench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "chain or golden" 2>&1 | tail -4 | tee gpurun_out/k_tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/k_bench.err < /dev/null | tee gpurun_out/k_bench.json | python tools/bench_brief.py
