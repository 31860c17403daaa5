#!/bin/bash
# PMC pass (counters only, with kernel-trace) over the per-op microbench; summaries -> gpurun_out/pmc_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run_pass() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python "$R/tools/microbench.py" --reps 3 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" > "$R/gpurun_out/pmc_$name.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel".ljust(72), *[n[-22:].rjust(24) for n in names])
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(k.ljust(72), *[("%.4g" % (d[n] / max(cnt[(k, n)], 1))).rjust(24) for n in names])
PY
  else echo "no counter file for $name"; tail -5 /tmp/pmc_$name.log; fi
}
if [ -n "$ONLY_TRAFFIC" ]; then
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
head -16 "$R/gpurun_out/pmc_fetch.txt" | cut -c1-140; head -16 "$R/gpurun_out/pmc_write.txt" | cut -c1-140
exit 0
fi
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pass sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
head -40 "$R/gpurun_out/pmc_sq.txt"; head -40 "$R/gpurun_out/pmc_sq2.txt"
