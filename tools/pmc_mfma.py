#!/usr/bin/env python3
"""MFMA utilisation per kernel from ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, plus GRBM_GUI_ACTIVE when it was
collected): util = MFMA-busy cycles summed over the chip's 1024 SIMDs / (1024 x the kernel's cycles).  The kernel's cycles
come from its own start / end timestamps at the nominal 2.4 GHz (the figure every `frac` in DESIGN.md is quoted against)
and, when GRBM_GUI_ACTIVE is there, from that counter (it comes summed over the 8 XCDs: / 8 = the cycles the GPU really spent;
the sustained clock under MFMA load is 2.05-2.4 GHz).
usage: pmc_mfma.py <counter_collection.csv> <out.csv> [name filter]"""
import csv
import sys
from collections import defaultdict

src, dst = sys.argv[1:3]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
SIMDS, GHZ, XCDS = 256 * 4, 2.4, 8
rows = defaultdict(dict)
for r in csv.DictReader(open(src)):
    d = rows[r["Dispatch_Id"]]
    d["name"] = r["Kernel_Name"]
    d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d in rows.values():
    if flt and flt not in d["name"]:
        continue
    a = agg[d["name"]]
    a[0] += 1
    a[1] += d["ns"]
    a[2] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    a[3] += d.get("GRBM_GUI_ACTIVE", 0.0)
w = csv.writer(open(dst, "w"))
w.writerow(["Kernel_Name", "Launches", "avg_us", "MFMA_busy_cycles_per_launch", "MfmaUtil_pct_of_2.4GHz_peak",
            "MfmaUtil_pct_of_GRBM_GUI_ACTIVE_per_XCD"])
for k in sorted(agg, key=lambda k: -agg[k][2]):
    n, ns, busy, gui = agg[k]
    if busy <= 0:
        continue
    u1 = 100.0 * busy / (SIMDS * ns * GHZ)
    u2 = 100.0 * busy / (SIMDS * gui / XCDS) if gui > 0 else float("nan")
    w.writerow([k, n, round(ns / n / 1e3, 2), round(busy / n, 0), round(u1, 2), round(u2, 2)])
    print(f"{k[:80]:80s} n={n:4d} avg={ns / n / 1e3:8.1f} us  MFMA util {u1:5.1f} % of the 2.4 GHz peak, {u2:5.1f} % of the active cycles")
