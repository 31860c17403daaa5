#!/bin/bash
# Round evidence in one GPU call: bench lines of every config, the drop-in loop, rocprofv3 kernel stats of the default bench, PMC traffic.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
timeout 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err < /dev/null
timeout 200 python bench.py --eager --no-cpu-baseline > gpurun_out/bench_eager.json 2>> gpurun_out/bench.err < /dev/null
for c in cfg2 cfg3; do
  timeout 200 python bench.py --config $c --steps 40 --warmup 8 > gpurun_out/bench_$c.json 2>> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --config $c --graph --steps 40 --warmup 8 > gpurun_out/bench_${c}_graph.json 2>> gpurun_out/bench.err < /dev/null
done
timeout 200 python bench.py --config cfg4 --steps 10 --warmup 2 > gpurun_out/bench_cfg4.json 2>> gpurun_out/bench.err < /dev/null
timeout 200 python bench.py --config cfg5 > gpurun_out/bench_cfg5.json 2>> gpurun_out/bench.err < /dev/null
timeout 200 python tools/bench_dropin.py > gpurun_out/dropin.txt 2>> gpurun_out/bench.err < /dev/null
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o trace -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err" < /dev/null)
f=$(find /tmp/prof_b -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv
if [ -z "$NO_PMC" ]; then
  ONLY_TRAFFIC=1 timeout 600 bash tools/pmc_run.sh > gpurun_out/pmc_traffic.log 2>&1 < /dev/null
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_m && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o trace -- python "$R/tools/microbench.py" --reps 3 > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_m -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/microbench_kernel_stats.csv
fi
for j in bench bench_eager bench_cfg2 bench_cfg2_graph bench_cfg3 bench_cfg3_graph bench_cfg4 bench_cfg5; do
  python - "$j" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "%.4g" % d["value"], d["unit"], "%.3f ms/step" % d["ms_per_step"], (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat gpurun_out/dropin.txt | tail -5; tail -3 gpurun_out/bench.err
