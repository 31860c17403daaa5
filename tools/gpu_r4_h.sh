#!/bin/bash
# pre-gathered gx / gy for the training chain (A/B), faster weight-preparation kernel
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for pg in 1 0; do
  echo "=== kbench DN_CHAIN_PREGATHER=$pg (16 x 10k)"; DN_CHAIN_PREGATHER=$pg timeout 120 ./tools/kbench --ops block_inf,block_fwd,block_bwd --reps 30 2>&1 | grep -v "^#" | cut -c1-60
done 2>&1 | tee gpurun_out/h_kbench.txt
for pg in 1 0; do
  DN_CHAIN_PREGATHER=$pg timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/h_bench_pg$pg.json 2> gpurun_out/h_bench.err < /dev/null
  python tools/bench_brief.py < gpurun_out/h_bench_pg$pg.json
done
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_h && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o trace -- "$R/tools/kbench" --ops block_fwd,block_bwd --reps 10 > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/prof_h -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | head -24
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained or golden or headline" 2>&1 | tail -2
