#!/bin/bash
# ONE parameterised GPU script (replaces the per-experiment tools/gpu_r*.sh of rounds 3-4): gpurun -- 'bash tools/gpu_run.sh <section> ...'.
# Every section writes under gpurun_out/ and prints a short summary.  Sections:
#   df_sweep     kbench of the diffusion operator: three launches vs the one-launch kernel for 1-4 mesh groups, both schedules, deferred /
#                immediate arrivals; forward + backward, fp64 spot checks
#   df_trace     s_memtime phase timeline of the one-launch kernel (libdiffnet_hip_dftrace.so = make variant TAG=dftrace EXTRA=-DDN_DF_TRACE=1)
#   df_small     the same comparison on one 7k-vertex mesh (BASELINE config 2) and on 64 x 2k meshes
#   c256         BASELINE config 4's width: kbench block tables chained / unfused at C = 256, the C = 256 parity cases, bench.py --config cfg4
#   kbench       block_inf / block_fwd / block_bwd / diffusion tables (tools/kbench --check)
#   tests [k]    GPU parity tier (optionally -k <expr>)
#   bench [args] bench.py (default flags) -> gpurun_out/bench.json
#   prof         rocprofv3 --kernel-trace --stats of the default bench -> gpurun_out/prof/
#   suite        tests/run_gpu_suite.sh (tests group by group, smoke, bench, rocprof)
#   evidence     the round's evidence in one call: GPU tier in one process (parity margins -> JSON), default bench line, eager / chain-off /
#                three-launch-diffusion variants, rocprofv3 kernel stats + per-step kernel listing, cfg2 kernel stats, kbench tables and,
#                unless NO_PMC is set, the FETCH_SIZE / WRITE_SIZE passes and the SQ passes of the block kernels
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
KB="timeout 180 ./tools/kbench"
sec=$1; shift
case "$sec" in
df_sweep)
  { echo "== three launches"; $KB --ops diffusion,diffusion_bwd --check --reps 30 --no-plan | cut -c1-170
    for g in 1 2 3 4; do for o in 0 1; do for f in 1 0; do
      echo "== one launch: groups $g order $o flags $f"; $KB --ops diffusion,diffusion_bwd --check --reps 30 --groups $g --opt diffuse_order=$o --opt diffuse_flags=$f | grep -v "^#" | cut -c1-170
    done; done; done; } 2>&1 | tee gpurun_out/df_sweep.txt ;;
df_trace)
  for g in ${1:-1 2 3}; do echo "== trace groups $g"; $KB --lib diffusion-net_amd/diffusion_net/libdiffnet_hip_dftrace.so --ops diffusion,diffusion_bwd --trace --reps 10 --groups $g | grep -v "^#" | cut -c1-260; done 2>&1 | tee gpurun_out/df_trace.txt ;;
df_modes)
  { for shape in "--meshes 16 --verts 10000" "--meshes 1 --verts 7000" "--meshes 64 --verts 2000" "--meshes 1 --verts 160000"; do
      for d in 0 2 1; do echo "== $shape: option diffuse=$d (0 row-GEMM back-projection, 2 direct back-projection launch, 1 one-launch kernel)"
        $KB $shape --ops diffusion,diffusion_bwd --check --reps 40 --groups 1 --opt diffuse=$d | grep -v "^#" | cut -c1-170; done
    done
    for d in 0 2; do echo "== blocks, 16 x 10k, diffuse=$d"; $KB --ops block_inf,block_fwd,block_bwd --reps 30 --groups 1 --opt diffuse=$d | grep -v "^#" | cut -c1-120; done
    for d in 0 2; do echo "== blocks, 1 x 7k, diffuse=$d"; $KB --meshes 1 --verts 7000 --ops block_inf,block_fwd,block_bwd --reps 50 --groups 1 --opt diffuse=$d | grep -v "^#" | cut -c1-120; done
    echo "== back-projection kernel phase stamps (cycles: planes staged, rows done)"; $KB --lib diffusion-net_amd/diffusion_net/libdiffnet_hip_dftrace.so --ops diffusion,diffusion_bwd --trace --reps 5 --groups 1 | grep -v "^#" | cut -c1-60
  } 2>&1 | tee gpurun_out/df_modes.txt ;;
df_small)
  { for shape in "--meshes 1 --verts 7000" "--meshes 64 --verts 2000" "--meshes 4 --verts 40000"; do
      echo "== $shape: three launches"; $KB $shape --ops diffusion,diffusion_bwd --check --reps 50 --no-plan | grep -v "^#" | cut -c1-170
      for g in 1 2 3; do echo "== $shape: one launch, groups $g"; $KB $shape --ops diffusion,diffusion_bwd --check --reps 50 --groups $g | grep -v "^#" | cut -c1-170; done
    done; } 2>&1 | tee gpurun_out/df_small.txt ;;
c256)
  { for c in 1 0; do echo "== 1 x 200k, C = K = 256, chain=$c"; $KB --meshes 1 --verts 200000 --C 256 --K 256 --ops block_inf,block_fwd --check --reps 10 --opt chain=$c | grep -v "^#" | cut -c1-170; done
    for c in 1 0; do echo "== 16 x 10k, C = 256, K = 128, chain=$c"; $KB --C 256 --ops block_inf,block_fwd --reps 10 --opt chain=$c | grep -v "^#" | cut -c1-170; done
  } 2>&1 | tee gpurun_out/c256_kbench.txt
  if [ -z "$NO_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "chained_forward_kernel_vs_unfused or chain_probes or large_inference" 2>&1 | tail -15 | tee gpurun_out/c256_tests.txt; fi
  timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-other-configs > gpurun_out/c256_bench_cfg4.json 2> gpurun_out/c256_bench.err; cat gpurun_out/c256_bench_cfg4.json | cut -c1-600; tail -3 gpurun_out/c256_bench.err ;;
sg256)
  { for o in 2 0; do echo "== 1 x 200k, C = K = 256, spectral_grad=$o"; $KB --meshes 1 --verts 200000 --C 256 --K 256 --ops block_inf,block_fwd --check --reps 10 --opt spectral_grad=$o | grep -v "^#" | cut -c1-170; done
    for o in 2 0; do echo "== kernel stats of block_inf, spectral_grad=$o"
      (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o trace -- "$R/tools/kbench" --lib "$R/diffusion-net_amd/diffusion_net/libdiffnet_hip.so" --meshes 1 --verts 200000 --C 256 --K 256 --ops block_inf --reps 10 --opt spectral_grad=$o > /tmp/prof_s.log 2>&1 < /dev/null)
      f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | grep -v "rocclr\|sg_grad\|sg_pack" | head -8; done
    for o in 2 1 0; do echo "== bench cfg4, spectral_grad=$o"; timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-other-configs --lib-opt spectral_grad=$o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
  } 2>&1 | tee gpurun_out/sg256.txt
  if [ -z "$NO_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "spectral_gradient or large_inference" 2>&1 | tail -15 | tee gpurun_out/sg256_tests.txt; fi ;;
bw256)
  { for d in 2 0; do echo "== 1 x 200k, C = K = 256, diffuse=$d (2: ring back-projection kernel, 0: row GEMM)"; $KB --meshes 1 --verts 200000 --C 256 --K 256 --ops diffusion,block_inf --check --reps 10 --opt diffuse=$d | grep -v "^#" | cut -c1-170; done
    for shape in "--meshes 16 --verts 10000" "--meshes 3 --verts 70000"; do for d in 2 0; do echo "== $shape, C = K = 256, diffuse=$d"; $KB $shape --C 256 --K 256 --ops diffusion,block_inf --check --reps 10 --opt diffuse=$d | grep -v "^#" | cut -c1-170; done; done
    echo "== kernel stats of block_inf, 1 x 200k"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_s && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o trace -- "$R/tools/kbench" --lib "$R/diffusion-net_amd/diffusion_net/libdiffnet_hip.so" --meshes 1 --verts 200000 --C 256 --K 256 --ops block_inf --reps 10 > /tmp/prof_s.log 2>&1 < /dev/null)
    f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | grep -v "rocclr\|sg_grad\|sg_pack" | head -9
    for d in 2 0; do echo "== bench cfg4, diffuse=$d"; timeout 300 python bench.py --config cfg4 --no-cpu-baseline --no-other-configs --lib-opt diffuse=$d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
  } 2>&1 | tee gpurun_out/bw256.txt
  if [ -z "$NO_TESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "backproject_wide or large_inference or ops" 2>&1 | tail -15 | tee gpurun_out/bw256_tests.txt; fi ;;
kbench)
  $KB --check "$@" 2>&1 | cut -c1-200 | tee gpurun_out/kbench.txt ;;
tests)
  if [ -n "$1" ]; then timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "$1" 2>&1 | tail -40 | tee gpurun_out/tests.txt
  else timeout 2400 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -40 | tee gpurun_out/tests.txt; fi ;;
bench)
  timeout 900 python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 4000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
prof)
  (cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -o trace -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err")
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-200 "$f" | head -40 ;;
suite)
  bash tests/run_gpu_suite.sh ;;
evidence)
  rm -f gpurun_out/parity_margins_cuda.json
  timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_tests_one_process.log 2>&1 < /dev/null; tail -3 gpurun_out/gpu_tests_one_process.log
  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --eager --no-cpu-baseline --no-other-configs > gpurun_out/bench_eager.json 2>> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --lib-opt chain=0 --no-cpu-baseline --no-other-configs > gpurun_out/bench_chain_off.json 2>> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --lib-opt diffuse=0 --no-cpu-baseline --no-other-configs > gpurun_out/bench_diffuse_off.json 2>> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --lib-opt spectral_grad=2 --no-cpu-baseline --no-other-configs > gpurun_out/bench_spectral_always.json 2>> gpurun_out/bench.err < /dev/null
  timeout 200 python bench.py --lib-opt spectral_grad=0 --no-cpu-baseline --no-other-configs > gpurun_out/bench_spectral_off.json 2>> gpurun_out/bench.err < /dev/null
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_b && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o trace -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > "$R/gpurun_out/prof_bench.json" 2> "$R/gpurun_out/prof.err" < /dev/null)
  f=$(find /tmp/prof_b -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/bench_kernel_stats.csv
  f=$(find /tmp/prof_b -name "*kernel_trace.csv" 2>/dev/null | head -1); [ -n "$f" ] && python tools/step_kernels.py "$f" > gpurun_out/step_kernels.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c2 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o trace -- python "$R/bench.py" --config cfg2 --steps 40 > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_c2 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/cfg2_kernel_stats.csv
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c4 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o trace -- python "$R/bench.py" --config cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_c4 -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/cfg4_kernel_stats.csv
  timeout 300 ./tools/kbench --check > gpurun_out/kbench.txt 2>&1 < /dev/null
  { for c in 1 0; do echo "== 1 x 200k, C = K = 256, chain=$c"; timeout 120 ./tools/kbench --meshes 1 --verts 200000 --C 256 --K 256 --ops block_inf,block_fwd --check --reps 10 --opt chain=$c | grep -v "^#" | cut -c1-170; done; } > gpurun_out/kbench_c256.txt 2>&1 < /dev/null
  timeout 300 ./tools/kbench --opt chain=0 --opt diffuse=0 --ops diffusion,diffusion_bwd,block_inf,block_fwd,block_bwd > gpurun_out/kbench_unfused.txt 2>&1 < /dev/null
  if [ -z "$NO_PMC" ]; then
    ONLY_TRAFFIC=1 timeout 600 bash tools/pmc_run.sh > gpurun_out/pmc_traffic.log 2>&1 < /dev/null
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_m && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o trace -- python "$R/tools/microbench.py" --reps 3 > /dev/null 2>&1 < /dev/null)
    f=$(find /tmp/prof_m -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/microbench_kernel_stats.csv
    [ -f gpurun_out/pmc_fetch.txt ] && python tools/traffic_summary.py gpurun_out/pmc_fetch.txt gpurun_out/pmc_write.txt gpurun_out/microbench_kernel_stats.csv gpurun_out/traffic.json > gpurun_out/traffic_summary.log 2>&1
    OPS=diffusion,diffusion_bwd,block_fwd,block_bwd,block_inf TAG=r06 timeout 600 bash tools/pmc_kbench.sh > gpurun_out/pmc_sq.log 2>&1 < /dev/null
  fi
  for j in bench bench_eager bench_chain_off bench_diffuse_off bench_spectral_always bench_spectral_off; do python tools/bench_brief.py < gpurun_out/$j.json; done
  tail -3 gpurun_out/bench.err; cat gpurun_out/traffic_summary.log 2>/dev/null | tail -8 ;;
*) echo "unknown section $sec"; exit 1 ;;
esac
