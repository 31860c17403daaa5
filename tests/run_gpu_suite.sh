#!/bin/bash
# Runs the GPU tier group by group (separate processes: a GPU fault in one group cannot mask the others),
# then the benchmark and a rocprofv3 kernel trace.  Outputs under gpurun_out/.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
rocminfo 2>/dev/null | grep -m1 gfx > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
: > gpurun_out/gpu_tests.log
for t in test_native_library_is_the_loaded_one test_golden_vectors test_single_ops test_one_launch_diffusion test_ragged_batches test_train_mode_dropout_masks_headline_width test_chain_probes_against_the_oracle \
         test_rna_like_wide_head test_nll_loss test_fused_head test_torch_compile_packed_forward test_real_mesh_pipeline test_device_packing_and_operator_cache test_autograph_reference_loop test_mismatched_patterns test_bitwise_determinism test_run_to_run_determinism_stress test_hks_and_label_smoothing test_gradient_sinks_accumulate_into_flat_bucket test_rccl_bucketed_all_reduce_world_size_one test_graph_captured_train_step test_graph_captures_the_rccl_gradient_all_reduce test_bench_two_ranks_on_one_gpu_over_gloo test_inkernel_dropout_matches_explicit_masks test_headline_shape_against_fp32_and_fp64_oracle \
         test_large_inference_shape test_size_independent_properties_at_full_size; do
  echo "=== $t" >> gpurun_out/gpu_tests.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "$t" 2>&1 | tail -40 >> gpurun_out/gpu_tests.log
done
grep -E "^===|passed|failed|error" gpurun_out/gpu_tests.log
echo "=== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
if [ -z "$NO_MICRO" ]; then echo "=== microbench"; timeout 600 python tools/microbench.py 2>&1 | grep -v Warn | tee gpurun_out/microbench.txt; fi
echo "=== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
  echo "=== rocprof"
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err")
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -30
fi
