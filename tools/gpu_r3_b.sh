#!/bin/bash
# round 3, GPU call B: automatic graph replay behind the reference signature + first/last linear magnitude words.
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
: > gpurun_out/b_tests.log
for t in test_autograph_reference_loop test_device_packing_and_operator_cache test_golden_vectors test_graph_captured_train_step test_bitwise_determinism test_headline_shape_against_fp32_and_fp64_oracle; do
  echo "=== $t" >> gpurun_out/b_tests.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "$t" 2>&1 | tail -30 >> gpurun_out/b_tests.log
done
grep -E "^===|passed|failed|error|Error" gpurun_out/b_tests.log
echo "=== dropin (autograph on / off)"
timeout 300 python tools/bench_dropin.py 2>&1 | grep -v Warn | tee gpurun_out/b_dropin.txt
DN_AUTOGRAPH=0 timeout 300 python tools/bench_dropin.py 2>&1 | grep -v Warn | tee -a gpurun_out/b_dropin.txt
echo "=== cfg2 (autograph on / off)"
timeout 300 python bench.py --config cfg2 --steps 40 > gpurun_out/b_cfg2.json 2> gpurun_out/b_cfg2.err; cut -c1-400 gpurun_out/b_cfg2.json; tail -3 gpurun_out/b_cfg2.err
DN_AUTOGRAPH=0 timeout 300 python bench.py --config cfg2 --steps 40 > gpurun_out/b_cfg2_off.json 2> gpurun_out/b_cfg2_off.err; cut -c1-400 gpurun_out/b_cfg2_off.json
echo "=== headline (no cpu baseline, no other configs)"
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; cut -c1-600 gpurun_out/b_bench.json; tail -3 gpurun_out/b_bench.err
timeout 600 python bench.py --eager --no-cpu-baseline --no-other-configs > gpurun_out/b_bench_eager.json 2>> gpurun_out/b_bench.err; cut -c1-300 gpurun_out/b_bench_eager.json
python - <<'PY'
import json
for f in ("b_bench.json", "b_bench_eager.json"):
    try:
        d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
        for k in d.get("kernel_families", []):
            print("   ", k.get("family"), k.get("launches_per_step"), k.get("avg_us"))
    except Exception as e:
        print(f, "unreadable", e)
PY
