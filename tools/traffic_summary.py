#!/usr/bin/env python
"""Combine the FETCH_SIZE / WRITE_SIZE PMC passes (tools/pmc_run.sh, ONLY_TRAFFIC=1) with the kernel-trace call counts
into HBM bytes per launch for each kernel family bench.py reports.  Corrections per MI355X_MICROARCH.md (HBM section):
both counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> x2."""
import csv
import json
import re
import sys


def read_pmc(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(.*?)\s{2,}([0-9.e+]+)\s*$", line)
        if m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def family(name):
    if name.startswith("void rowgemm_persist_kernel") or name.startswith("void rowgemm_ws_kernel"):
        return "rowgemm_kernel<*,1>"
    if name.startswith("void rowgemm_ws2_kernel"):
        return "rowgemm_kernel<*,2>"
    m = re.match(r"void rowgemm_kernel<\d+, \d+, \d+, (\d)", name)
    if m:
        return "rowgemm_kernel<*,%s>" % m.group(1)
    if "tngemm_x3_multi_kernel" in name:           # (one key per kernel since round 5: bench.py's brackets are per kernel)
        return "tngemm_x3_multi_kernel"
    if "tngemm_da_kernel" in name:
        return "tngemm_da_kernel"
    if "tngemm_kernel" in name or "tngemm_x3_kernel" in name:
        return "tngemm_kernel"
    if "backproject_kernel" in name:
        return "backproject_kernel"
    if "diffuse_kernel" in name:
        return "diffuse_kernel"
    if "spmm_kernel" in name:
        return "spmm_kernel"
    if "chain_fwd_kernel" in name:
        return "chain_fwd_kernel"
    if "chain_bwd_kernel" in name:
        return "chain_bwd_kernel"
    return None


def main(fetch_txt, write_txt, stats_csv, out_json):
    fetch, write = read_pmc(fetch_txt), read_pmc(write_txt)
    fam = {}
    for r in csv.DictReader(open(stats_csv)):
        f = family(r["Name"])
        key = r["Name"][:70]
        if f is None or key not in fetch or key not in write:
            continue
        calls = int(r["Calls"])
        by = (2.0 * fetch[key] + write[key]) * 1024.0
        d = fam.setdefault(f, {"calls": 0, "bytes": 0.0, "members": {}})
        d["calls"] += calls
        d["bytes"] += calls * by
        d["members"][key] = {"calls": calls, "fetch_KiB_raw": fetch[key], "write_KiB": write[key], "hbm_bytes_per_launch": by}
    res = {f: {"hbm_bytes_per_launch": d["bytes"] / d["calls"], "members": d["members"]} for f, d in fam.items()}
    json.dump(res, open(out_json, "w"), indent=1)
    for f, d in res.items():
        print(f, "%.1f MB/launch" % (d["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
