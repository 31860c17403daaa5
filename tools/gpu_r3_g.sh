#!/bin/bash
# per-kernel times of the unmodified one-mesh-per-step loop (cfg2, automatic graph replay) for the small-mesh analysis
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c2 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o trace -- python "$R/bench.py" --config cfg2 --steps 40 > "$R/gpurun_out/g_cfg2.json" 2> "$R/gpurun_out/g_cfg2.err" < /dev/null)
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/g_cfg2_kernel_stats.csv
cut -c1-200 gpurun_out/g_cfg2.json | tail -1; head -5 gpurun_out/g_cfg2_kernel_stats.csv | cut -c1-150
