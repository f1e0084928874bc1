#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 --pmc counter_collection.csv.
usage: pmc_summary.py <counter_collection.csv> <COUNTER> <out.csv>"""
import csv
import sys
from collections import defaultdict

src, counter, dst = sys.argv[1:4]
tot, n = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(src)):
    if r["Counter_Name"] == counter:
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]] += 1
w = csv.writer(open(dst, "w"))
w.writerow(["Kernel_Name", "Launches", f"{counter}_KiB_total", f"{counter}_KiB_per_launch"])
for k in sorted(tot, key=lambda k: -tot[k]):
    w.writerow([k, n[k], round(tot[k], 1), round(tot[k] / n[k], 2)])
