#!/bin/bash
# chunks per workgroup in the batched weight-gradient launch (fewer partials vs fewer workgroups)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for bg in 1 2 3 4; do echo "=== DN_TN_BATCH_GROUP=$bg"; DN_TN_BATCH_GROUP=$bg timeout 120 ./tools/kbench --ops block_bwd --reps 40 2>&1 | grep -v "^#" | cut -c1-60; done | tee gpurun_out/j_kbench.txt
for bg in 1 2 3; do DN_TN_BATCH_GROUP=$bg timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2> gpurun_out/j_bench.err < /dev/null | python tools/bench_brief.py; done
