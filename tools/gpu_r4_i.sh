#!/bin/bash
# one launch for the three weight-gradient products of a block (chained-backward path), faster prep kernel
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 120 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 30 2>&1 | grep -v "^#" | cut -c1-60 | tee gpurun_out/i_kbench.txt
DN_CHAIN_MIN_ROWS=0 timeout 120 ./tools/kbench --meshes 1 --verts 7000 --ops block_fwd,block_bwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60 | tee -a gpurun_out/i_kbench.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err < /dev/null; python tools/bench_brief.py < gpurun_out/i_bench.json
timeout 300 python bench.py --config cfg2 --steps 40 2>> gpurun_out/i_bench.err | python tools/bench_brief.py
timeout 300 python bench.py --config cfg2 --graph --steps 40 2>> gpurun_out/i_bench.err | python tools/bench_brief.py
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or golden or headline or ragged or determinism or wide_head or graph" 2>&1 | tail -2
