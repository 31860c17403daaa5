#!/usr/bin/env python3
"""List the kernels of ONE replayed train step from a rocprofv3 --kernel-trace CSV: the launches between the last two optimizer
kernels, in start order, with durations and the idle gap before each.  usage: step_kernels.py <kernel_trace.csv> [marker-substring]"""
import csv, re, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:86]

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "FusedAdam"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("marker kernel seen fewer than twice")
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t_prev = int(rows[a]["End_Timestamp"])
busy = gap = 0
print(f"{len(step)} launches between the last two '{marker}' kernels")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = max(0, s - t_prev)
    busy += e - s; gap += g
    print(f"{(e - s) / 1e3:9.1f} us  gap {g / 1e3:6.1f}  {short(r['Kernel_Name'])}")
    t_prev = max(t_prev, e)
print(f"busy {busy / 1e3:.1f} us, idle between kernels {gap / 1e3:.1f} us, wall {(busy + gap) / 1e3:.1f} us")
