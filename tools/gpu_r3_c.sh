#!/bin/bash
# round 3, GPU call: autograph test with full output, probe of the live-graph case, cfg2 on/off
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 120 python tools/graph_probe.py ag_live_graph 2>&1 | grep -v amdgpu.ids | tail -3
DN_PARITY_VERBOSE=1 timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -s --tb=short -k "test_autograph_reference_loop or test_device_packing_and_operator_cache" > gpurun_out/c_autograph.log 2>&1
grep -vE "^  File|Extension modules" gpurun_out/c_autograph.log | cut -c1-300 | tail -40
