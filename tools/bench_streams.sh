#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for s in 1 2 4 1 2; do echo "=== streams=$s"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.3fM verts/s  ms/step %.2f' % (d['value']/1e6, d['ms_per_step'])); [print('   ', f['kernel'], '%.1f ms total, avg %.1f us, %.1f TF' % (f['ms_total'], f['avg_us'], f['tflops'])) for f in d['kernel_families']]"; done | tee gpurun_out/streams.txt
