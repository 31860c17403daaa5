#!/bin/bash
# round 3, GPU call C: autograph test with full output, cfg2 on/off
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
DN_PARITY_VERBOSE=1 timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -s --tb=short -k "test_autograph_reference_loop or test_device_packing_and_operator_cache" > gpurun_out/c_autograph.log 2>&1
grep -vE "^  File|Extension modules" gpurun_out/c_autograph.log | cut -c1-300 | tail -60
echo "=== cfg2 (autograph on / off)"
timeout 300 python bench.py --config cfg2 --steps 40 > gpurun_out/c_cfg2.json 2> gpurun_out/c_cfg2.err; cut -c1-300 gpurun_out/c_cfg2.json; tail -3 gpurun_out/c_cfg2.err
DN_AUTOGRAPH=0 timeout 300 python bench.py --config cfg2 --steps 40 > gpurun_out/c_cfg2_off.json 2> gpurun_out/c_cfg2_off.err; cut -c1-300 gpurun_out/c_cfg2_off.json
