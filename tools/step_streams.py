#!/usr/bin/env python3
"""One replayed training step (between two adamw launches) from a rocprofv3 --kernel-trace CSV, per HIP queue: every
kernel with its start offset, duration and the idle gap in front of it on ITS queue; the long gaps of the main queue are the
places where it waits for another stream (joins) or for the host.
usage: step_streams.py <dir-or-csv> [min_gap_us]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

path = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-12], idx[-11]          # a steady-state step of the timed region (the last 24 are the event-instrumented ones)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:46]


step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
byq = defaultdict(list)
for r in step:
    byq[r.get("Queue_Id")].append(r)
main = max(byq, key=lambda q: len(byq[q]))
print(f"step wall {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(step)} kernels on {len(byq)} queues (main = queue {main})")
for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3
    print(f"-- queue {q}: {len(ks)} kernels, busy {busy:.1f} us")
    prev = t0 if q == main else None
    for r in ks:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        flag = "  <== gap" if (q == main and gap >= min_gap) else ""
        if q != main or gap >= min_gap:
            print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short(r['Kernel_Name'])}{flag}")
        prev = e
    if q == main:
        gaps = []
        p = t0
        for r in ks:
            gaps.append((int(r["Start_Timestamp"]) - p) / 1e3)
            p = int(r["End_Timestamp"])
        print(f"   main queue: sum of gaps {sum(gaps):.1f} us, of which gaps >= {min_gap} us: {sum(g for g in gaps if g >= min_gap):.1f} us in {sum(1 for g in gaps if g >= min_gap)} places")
