#!/bin/bash
# the one-unit row kernel for small batches: kbench and cfg2 with it (default) and without (DN_ONE_MAX_TILES=0), parity tests
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for cfg in "--meshes 1 --verts 7000" "--meshes 2 --verts 16000" "--meshes 4 --verts 12000"; do
  for off in 0 ""; do
    echo "=== kbench $cfg DN_ONE_MAX_TILES='${off}'"
    DN_ONE_MAX_TILES=$off timeout 120 ./tools/kbench $cfg --ops block_inf,block_fwd,block_bwd --reps 50 --check 2>&1 | grep -v "^#" | cut -c1-120
  done
done 2>&1 | sed 's/DN_ONE_MAX_TILES=.. /&/' | tee gpurun_out/h_kbench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "golden or autograph or ragged or inkernel or determinism or single_ops or fused_head or compile" 2>&1 | tail -4
for off in 0 ""; do
  echo "=== cfg2 DN_ONE_MAX_TILES='${off}'"
  DN_ONE_MAX_TILES=$off timeout 300 python bench.py --config cfg2 --steps 40 2> gpurun_out/h.err | python tools/bench_brief.py
  DN_ONE_MAX_TILES=$off timeout 300 python bench.py --config cfg2 --graph --steps 40 2>> gpurun_out/h.err | python tools/bench_brief.py
done 2>&1 | tee gpurun_out/h_cfg2.txt
