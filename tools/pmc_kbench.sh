#!/bin/bash
# SQ counter passes (counters only + kernel trace, separate rocprofv3 runs) over tools/kbench ops; summaries -> gpurun_out/pmck_*.txt
#   OPS=from_basis,linear LIB=path.so [KARGS="--meshes 1 --verts 200000 --C 256 --K 256"] bash tools/pmc_kbench.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OPS=${OPS:-from_basis}
LIB=${LIB:-$R/diffusion-net_amd/diffusion_net/libdiffnet_hip.so}
TAG=${TAG:-k}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run_pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmck_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmck_$name -o p -- "$R/tools/kbench" --lib "$LIB" --reps 3 --ops "$OPS" $KARGS > /tmp/pmck_$name.log 2>&1
  f=$(find /tmp/pmck_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" > "$R/gpurun_out/pmck_${TAG}_$name.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel".ljust(62), *[n[-20:].rjust(21) for n in names])
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(k.ljust(62), *[("%.4g" % (d[n] / max(cnt[(k, n)], 1))).rjust(21) for n in names])
PY
  else echo "no counter file for $name"; tail -5 /tmp/pmck_$name.log; fi
}
run_pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run_pass sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_WAVES
cat "$R"/gpurun_out/pmck_${TAG}_sq*.txt | cut -c1-240
