#!/bin/bash
# interleaved A/B of library build variants on the per-op microbench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; mkdir -p gpurun_out
for round in ${ROUNDS:-1 2}; do for v in "" $VARIANTS; do echo "=== variant='$v' round $round"; DN_LIB_VARIANT=$v timeout 300 python tools/microbench.py --reps 20 2>&1 | grep -vE "Warn|amdgpu.ids"; done; done | tee gpurun_out/ab.txt
