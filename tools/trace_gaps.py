#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-queue busy time, gaps between consecutive kernels,
wall span.  Usage: trace_gaps.py <dir-or-csv> [skip_fraction]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]          # steady state only
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
span = (t1 - t0) / 1e3
print(f"{path}: {len(rows)} kernels, span {span:.1f} us")
byq = defaultdict(list)
for r in rows:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in byq.items():
    busy = sum(e - s for s, e, _ in ks) / 1e3
    gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
    gaps_s = sorted(gaps)
    n = len(gaps_s)
    print(f"queue {q}: {len(ks)} kernels, busy {busy:.1f} us ({100 * busy / span:.1f}% of span), "
          f"gap median {gaps_s[n // 2]:.2f} us, mean {sum(gaps) / max(n, 1):.2f}, p90 {gaps_s[int(n * .9)]:.2f}, "
          f"sum of gaps<50us {sum(g for g in gaps if g < 50):.1f} us")
# union busy
ev = sorted((s, e) for ks in byq.values() for s, e, _ in ks)
cur_s, cur_e, union = ev[0][0], ev[0][1], 0
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"GPU busy (union over queues): {union / 1e3:.1f} us = {100 * union / 1e3 / span:.1f}% of span")
