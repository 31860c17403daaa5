#!/bin/bash
# GPU tier with the new cases (chain vs unfused, GraphedEpoch, autograph output modes), parity margins, epoch mode, cfg3 with the C = 64 chain
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; rm -f gpurun_out/parity_margins.json
timeout 1200 python -m pytest tests/ -x -q -m gpu > gpurun_out/e_gpu_tests.log 2>&1 < /dev/null; tail -4 gpurun_out/e_gpu_tests.log
timeout 400 python bench.py --epoch 8 --steps 24 --warmup 8 2> gpurun_out/e_epoch.err | tee gpurun_out/e_epoch.json | python tools/bench_brief.py
for ch in 1 0; do
  DN_CHAIN=$ch timeout 300 python bench.py --config cfg3 --graph --steps 20 2>> gpurun_out/e_bench.err | python tools/bench_brief.py
done
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/e_bench.json 2>> gpurun_out/e_bench.err; python tools/bench_brief.py < gpurun_out/e_bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/e_bench.json')); print(json.dumps(d.get('parity'), indent=None)[:600])
PY
