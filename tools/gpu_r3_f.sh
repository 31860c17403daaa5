#!/bin/bash
# clock / power while one kernel loops: is the row GEMM power-limited?  (rocm-smi sampled every 0.2 s during ~3 s of back-to-back launches)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
L=diffusion-net_amd/diffusion_net
sample() { for i in $(seq 1 ${1:-10}); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.2; done; }
echo "=== idle"; sample 3
for v in "" _ko8 _ko7 _ko1 _ko4; do
  for op in linear copyk; do
    [ "$op" = copyk ] && [ -n "$v" ] && continue
    echo "=== variant '${v}' op $op"
    ( timeout 60 ./tools/kbench --lib $L/libdiffnet_hip${v}.so --ops $op --reps ${REPS:-40000} 2>&1 | grep -v "^#" | cut -c1-60 | head -4 ) &
    sleep 1.2; sample 6; wait
  done
done 2>&1 | tee gpurun_out/f_power.txt
