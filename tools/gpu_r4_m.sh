#!/bin/bash
# block backward with the parameter path on a helper stream: kernel times, parity tier, bench with and without
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for ov in 0 1; do echo "=== DN_BWD_OVERLAP=$ov"; DN_BWD_OVERLAP=$ov timeout 200 ./tools/kbench --ops block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-70; done | tee gpurun_out/m_kbench.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/m_tests.txt
for ov in 0 1; do DN_BWD_OVERLAP=$ov timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/m_bench$ov.err < /dev/null | tee gpurun_out/m_bench$ov.json | python tools/bench_brief.py; done
DN_BWD_OVERLAP=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --config cfg2 --graph 2>> gpurun_out/m_bench1.err < /dev/null | python tools/bench_brief.py
DN_BWD_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --config cfg2 --graph 2>> gpurun_out/m_bench0.err < /dev/null | python tools/bench_brief.py
