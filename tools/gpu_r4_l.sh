#!/bin/bash
# kernel sequence of one replayed headline step (and of one cfg2 step)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for cfg in headline cfg2; do
  extra=""; [ $cfg = cfg2 ] && extra="--config cfg2 --graph"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_l && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -o trace -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs $extra > /dev/null 2>&1 < /dev/null)
  f=$(find /tmp/prof_l -name "*kernel_trace.csv" | head -1)
  python tools/step_kernels.py "$f" > gpurun_out/l_step_$cfg.txt 2>&1
  tail -1 gpurun_out/l_step_$cfg.txt
done
