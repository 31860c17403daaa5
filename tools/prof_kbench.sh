#!/bin/bash
# rocprofv3 kernel trace + stats over tools/kbench ops; summary -> gpurun_out/kstats_<TAG>.csv
#   OPS=gradfeat_bwd TAG=da bash tools/prof_kbench.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OPS=${OPS:-all}
LIB=${LIB:-$R/diffusion-net_amd/diffusion_net/libdiffnet_hip.so}
TAG=${TAG:-k}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_$TAG
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$TAG -o t -- "$R/tools/kbench" --lib "$LIB" --ops "$OPS" > /tmp/kst_$TAG.log 2>&1 < /dev/null
f=$(find /tmp/kst_$TAG -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$R/gpurun_out/kstats_$TAG.csv"; cut -c1-150 "$f" | head -${LINES_OUT:-12}; else echo "no stats file"; tail -5 /tmp/kst_$TAG.log; fi
