#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into HBM bytes per
launch of EVERY kernel symbol, tagged with the hash of the sources they were measured on.

  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(on gfx950 FETCH_SIZE under-reports a wide coalesced read stream by 2x: MI355X_MICROARCH.md, HBM / rocprofv3 section; both
counters are in KiB.)  bench.py looks a roofline site's kernel symbol up in this file and REFUSES the file when its
`source_hash` is not the hash of the sources it is running (dostransformer_amd._lib.source_hash), so a stale file can
never be reported as `roofline.traffic`.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [git_head] [bench config]"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}, n


def short(sym):
    """'void (anonymous namespace)::gemm_kernel<3, 2, 0, 0, 1, 1>((anonymous namespace)::GemmLaunch)' -> 'gemm_kernel<3, 2, 0, 0, 1, 1>'"""
    s = sym.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(s):          # cut the argument list: first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i].strip()
    return s.strip()


def main():
    from dostransformer_amd._lib import source_hash
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    config = sys.argv[5] if len(sys.argv) > 5 else "phonon_h128_b64"
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --config " + config +
                    " --no-cpu-baseline --no-secondary` (the replayed step itself), averaged per launch of each kernel "
                    "symbol; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, "
                    "MI355X_MICROARCH.md HBM section).",
           "config": config,
           "source_hash": source_hash(),
           "git_head": sys.argv[4] if len(sys.argv) > 4 else "unknown",
           "kernels": {}}
    for k in sorted(fetch, key=lambda k: -fetch[k] * nf[k]):
        f, w = fetch[k], write.get(k, 0.0)
        out["kernels"][short(k)] = {"launches": nf[k], "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                                    "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(f"{len(out['kernels'])} kernel symbols, source_hash {out['source_hash']}")


if __name__ == "__main__":
    main()
