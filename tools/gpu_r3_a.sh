#!/bin/bash
# round 3, GPU call A: new correctness cases, HBM calibration with hand-written kernels, baseline kbench, SQ counters of the shipped kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -m1 gfx > gpurun_out/gpu.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -k "native or golden or operator_cache or fused_head or nll_loss or rna_like" 2>&1 | tail -15 > gpurun_out/a_tests.log
cat gpurun_out/a_tests.log | tail -5
timeout 300 ./tools/kbench --ops copyk > gpurun_out/a_copyk.txt 2>&1; cat gpurun_out/a_copyk.txt
timeout 300 ./tools/kbench --check > gpurun_out/a_kbench.txt 2>&1; cat gpurun_out/a_kbench.txt
OPS=diffusion,linear,gradfeat,gradfeat_bwd TAG=r03 timeout 600 bash tools/pmc_kbench.sh > gpurun_out/a_pmc.log 2>&1; tail -60 gpurun_out/a_pmc.log | cut -c1-250
