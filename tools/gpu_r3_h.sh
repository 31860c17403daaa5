#!/bin/bash
# workgroup-level magnitude commits (smallk + wave-specialised row GEMM): kbench at small sizes with the per-wave variant beside it, cfg2, parity
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
for cfg in "--meshes 1 --verts 7000" "--meshes 16 --verts 10000"; do
  for v in _wavecommit ""; do
    echo "=== kbench $cfg lib '${v}'"
    timeout 120 ./tools/kbench --lib $L/libdiffnet_hip${v}.so $cfg --ops block_inf,block_fwd,block_bwd --reps 50 2>&1 | grep -v "^#" | cut -c1-60
  done
done 2>&1 | tee gpurun_out/h_kbench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "golden or autograph or ragged or inkernel or determinism or single_ops or headline" 2>&1 | tail -3
for v in wavecommit ""; do
  echo "=== cfg2 lib variant '${v}'"
  DN_LIB_VARIANT=$v timeout 300 python bench.py --config cfg2 --steps 40 2> gpurun_out/h.err | python tools/bench_brief.py
  DN_LIB_VARIANT=$v timeout 300 python bench.py --config cfg2 --graph --steps 40 2>> gpurun_out/h.err | python tools/bench_brief.py
done 2>&1 | tee gpurun_out/h_cfg2.txt
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c2 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o trace -- python "$R/bench.py" --config cfg2 --steps 40 > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/h_cfg2_kernel_stats.csv && grep -E "smallk|rowgemm_ws_kernel<4|rowgemm_ws_kernel<1, true, true, 3" gpurun_out/h_cfg2_kernel_stats.csv | cut -c1-130
