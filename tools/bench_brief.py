"""Print the essentials of bench.py JSON lines read from stdin (development aid)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    fam = "  ".join("%s %dx%.1fus" % (f["kernel"].replace("rowgemm_kernel", "rg").replace("_kernel", ""), f["launches"], f["avg_us"]) for f in d.get("kernel_families", []))
    print("%.3f M %s  %.3f ms/step  roofline %.3f  diffusion %.3f | %s" % (d["value"] / 1e6, d["unit"], d["ms_per_step"], (d.get("roofline") or {}).get("frac", 0),
                                                                  (d.get("diffusion_block") or {}).get("frac_hbm_8TBs", 0), fam),
          "| roofline kernel", (d.get("roofline") or {}).get("kernel"), "| diffusion bwd frac", ((d.get("diffusion_block") or {}).get("backward") or {}).get("frac_hbm_8TBs"))
