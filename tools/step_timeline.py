#!/usr/bin/env python3
"""Print the kernels of ONE training step (between two adamw launches) from a rocprofv3
--kernel-trace CSV of an eager single-stream run: order, short name, workgroups, duration."""
import csv
import glob
import os
import re
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
tot = 0.0
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n
for i, r in enumerate(rows[a + 1:b + 1]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    print(f"{i:3d} {short(r['Kernel_Name']):58s} wgs={wg:5d} lds={r.get('LDS_Block_Size','?'):>6s} {d:7.1f} us")
print(f"sum {tot:.1f} us over {b - a} kernels; wall {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us")
