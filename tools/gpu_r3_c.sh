#!/bin/bash
# round 3, GPU call: autograph test (both sizes in one process) + graph tests after the workspace-under-capture change
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
DN_PARITY_VERBOSE=1 timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -s --tb=short -k "test_autograph_reference_loop or test_device_packing_and_operator_cache or test_graph_captured_train_step or test_graph_captures_the_rccl" > gpurun_out/c_autograph.log 2>&1
grep -vE "^  File|Extension modules|Warning|warn" gpurun_out/c_autograph.log | cut -c1-300 | tail -30
echo "=== cfg2 --graph (explicit whole-step graphs: 8 captures in one process)"
timeout 300 python bench.py --config cfg2 --graph --steps 40 > gpurun_out/c_cfg2_graph.json 2> gpurun_out/c_cfg2_graph.err; cut -c1-200 gpurun_out/c_cfg2_graph.json | tail -1; tail -2 gpurun_out/c_cfg2_graph.err
