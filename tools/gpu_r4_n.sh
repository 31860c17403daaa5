#!/bin/bash
# kernel sequence of one cfg2 step through the unmodified reference loop (autograph)
mkdir -p gpurun_out; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_n && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_n -o trace -- python "$R/bench.py" --steps 24 --warmup 16 --no-cpu-baseline --no-other-configs --config cfg2 > /dev/null 2>&1 < /dev/null)
f=$(find /tmp/prof_n -name "*kernel_trace.csv" | head -1)
python tools/step_kernels.py "$f" nll_loss_forward > gpurun_out/n_step_cfg2.txt 2>&1
tail -1 gpurun_out/n_step_cfg2.txt
