#!/bin/bash
# Round evidence in one GPU call: GPU test tier, the default bench line (incl. the other configs' sub-runs), eager / split-bf16-only variants,
# rocprofv3 kernel stats of the default bench, PMC traffic passes, kbench tables.  Everything lands in gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
# the driver's form first: the whole GPU tier in ONE process
timeout 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests_one_process.log 2>&1 < /dev/null; tail -3 gpurun_out/gpu_tests_one_process.log
if [ -z "$NO_GROUPS" ]; then NO_MICRO=1 NO_PROF=1 BENCH_STEPS=1 bash tests/run_gpu_suite.sh > gpurun_out/suite.log 2>&1; fi
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err < /dev/null
timeout 200 python bench.py --eager --no-cpu-baseline --no-other-configs > gpurun_out/bench_eager.json 2>> gpurun_out/bench.err < /dev/null
DN_F16=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/bench_bf16x3_only.json 2>> gpurun_out/bench.err < /dev/null
timeout 200 python tools/bench_dropin.py > gpurun_out/dropin.txt 2>> gpurun_out/bench.err < /dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o trace -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err" < /dev/null)
f=$(find /tmp/prof_b -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv
timeout 300 ./tools/kbench --check > gpurun_out/kbench.txt 2>&1 < /dev/null
DN_F16=0 timeout 300 ./tools/kbench --ops block_inf,block_fwd,block_bwd > gpurun_out/kbench_bf16x3_only.txt 2>&1 < /dev/null
if [ -z "$NO_PMC" ]; then
  ONLY_TRAFFIC=1 timeout 600 bash tools/pmc_run.sh > gpurun_out/pmc_traffic.log 2>&1 < /dev/null
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_m && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o trace -- python "$R/tools/microbench.py" --reps 3 > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_m -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/microbench_kernel_stats.csv
  [ -f gpurun_out/pmc_fetch.txt ] && python tools/traffic_summary.py gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt gpurun_out/microbench_kernel_stats.csv gpurun_out/traffic.json > gpurun_out/traffic_summary.log 2>&1
  OPS=linear_relu,from_basis,block_fwd TAG=r03f timeout 600 bash tools/pmc_kbench.sh > gpurun_out/pmc_sq.log 2>&1 < /dev/null
fi
grep -E "^===|passed|failed|error" gpurun_out/gpu_tests.log | tail -60
for j in bench bench_eager bench_bf16x3_only; do python tools/bench_brief.py < gpurun_out/$j.json; done
tail -5 gpurun_out/dropin.txt; tail -3 gpurun_out/bench.err; cat gpurun_out/traffic_summary.log 2>/dev/null | tail -8
