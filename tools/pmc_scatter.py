#!/usr/bin/env python3
"""Achieved HBM GB/s of the scatter-add kernel from the counters: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`tools/bench_kernels.py --what scatter`, matched launch by launch (same program, same launch order); HBM bytes of a launch =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of MI355X_MICROARCH.md, as in tools/pmc_traffic.py), rate = bytes /
the launch's duration in the FETCH pass.  Launches are grouped by grid size = the cases of the microbenchmark.
usage: pmc_scatter.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> [kernel name filter]"""
import csv
import sys
from collections import defaultdict

ff, fw, dst = sys.argv[1:4]
flt = sys.argv[4] if len(sys.argv) > 4 else "segment_reduce"


def launches(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and flt in r["Kernel_Name"]:
            out.append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"]),
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out.sort()
    return out


f, w = launches(ff, "FETCH_SIZE"), launches(fw, "WRITE_SIZE")
assert len(f) == len(w) and all(a[1] == b[1] for a, b in zip(f, w)), "the two passes do not line up"
g = defaultdict(lambda: [0, 0.0, 0.0])
for (_, grid, fk, ns), (_, _, wk, _) in zip(f, w):
    a = g[grid]
    a[0] += 1
    a[1] += (2.0 * fk + wk) * 1024.0
    a[2] += ns
wr = csv.writer(open(dst, "w"))
wr.writerow(["kernel", "grid_size", "launches", "avg_us", "HBM_MB_per_launch", "GB_per_s", "pct_of_8TBs"])
for grid in sorted(g):
    n, by, ns = g[grid]
    gbs = by / ns
    wr.writerow([flt, grid, n, round(ns / n / 1e3, 2), round(by / n / 1e6, 2), round(gbs, 1), round(100 * gbs / 8000, 1)])
    print(f"{flt} grid {grid:9d}: {n:4d} launches, {ns / n / 1e3:8.1f} us, {by / n / 1e6:9.2f} MB per launch, {gbs:7.1f} GB/s = {100 * gbs / 8000:5.1f} % of 8 TB/s")
