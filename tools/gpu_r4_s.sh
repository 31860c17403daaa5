#!/bin/bash
# clamp folded into the block's first launch, spectral backward in one launch: GPU tier, kernel times, launches per step
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/s_tests.txt
timeout 200 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70 | tee gpurun_out/s_kbench.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/s_bench.err < /dev/null | tee gpurun_out/s_bench.json | python tools/bench_brief.py
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_s -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python tools/step_kernels.py "$f" > gpurun_out/s_step_kernels.txt 2>&1; head -1 gpurun_out/s_step_kernels.txt; tail -1 gpurun_out/s_step_kernels.txt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --config cfg2 2>> gpurun_out/s_bench.err < /dev/null | python tools/bench_brief.py
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --config cfg2 --graph 2>> gpurun_out/s_bench.err < /dev/null | python tools/bench_brief.py
