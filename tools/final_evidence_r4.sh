#!/bin/bash
# Round-4 evidence in one GPU call: GPU test tier (one process, parity margins -> JSON), the default bench line (incl. the other configs' sub-runs and
# epoch mode), eager / chain-off variants, rocprofv3 kernel stats of the default bench and of cfg2, PMC traffic passes (FETCH_SIZE / WRITE_SIZE in their
# own passes), SQ passes of the chained kernels, kbench tables.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out; rm -f gpurun_out/parity_margins.json
timeout 1200 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests_one_process.log 2>&1 < /dev/null; tail -3 gpurun_out/gpu_tests_one_process.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err < /dev/null
timeout 200 python bench.py --eager --no-cpu-baseline --no-other-configs > gpurun_out/bench_eager.json 2>> gpurun_out/bench.err < /dev/null
DN_CHAIN=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/bench_chain_off.json 2>> gpurun_out/bench.err < /dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o trace -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err" < /dev/null)
f=$(find /tmp/prof_b -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv
f=$(find /tmp/prof_b -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/step_kernels.py "$f" > gpurun_out/step_kernels.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c2 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o trace -- python "$R/bench.py" --config cfg2 --steps 40 > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/cfg2_kernel_stats.csv
timeout 300 ./tools/kbench --check > gpurun_out/kbench.txt 2>&1 < /dev/null
DN_CHAIN=0 timeout 300 ./tools/kbench --ops block_inf,block_fwd,block_bwd > gpurun_out/kbench_chain_off.txt 2>&1 < /dev/null
if [ -z "$NO_PMC" ]; then
  ONLY_TRAFFIC=1 timeout 600 bash tools/pmc_run.sh > gpurun_out/pmc_traffic.log 2>&1 < /dev/null
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_m && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o trace -- python "$R/tools/microbench.py" --reps 3 > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_m -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/microbench_kernel_stats.csv
  [ -f gpurun_out/pmc_fetch.txt ] && python tools/traffic_summary.py gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt gpurun_out/microbench_kernel_stats.csv gpurun_out/traffic.json > gpurun_out/traffic_summary.log 2>&1
  OPS=block_fwd,block_bwd TAG=r04 timeout 600 bash tools/pmc_kbench.sh > gpurun_out/pmc_sq.log 2>&1 < /dev/null
fi
for j in bench bench_eager bench_chain_off; do python tools/bench_brief.py < gpurun_out/$j.json; done
tail -3 gpurun_out/bench.err; cat gpurun_out/traffic_summary.log 2>/dev/null | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
print("configs:", {k: (round(v.get("value", 0) / 1e6, 2) if isinstance(v, dict) and "value" in v else v) for k, v in d.get("configs", {}).items()})
print("epoch_mode:", {k: d.get("epoch_mode", {}).get(k) for k in ("value", "ratio_to_static_batch_replay")})
print("roofline:", {k: d["roofline"].get(k) for k in ("kernel", "bound", "achieved", "frac", "traffic", "avg_launch_us")})
print("diffusion_block:", {k: d["diffusion_block"].get(k) for k in ("ms", "gbps", "frac_hbm_8TBs")})
PY
