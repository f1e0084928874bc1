#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
HBM bytes per launch for the kernels behind bench.py's roofline sites.

  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(on gfx950 FETCH_SIZE under-reports a wide coalesced read stream by 2x: MI355X_MICROARCH.md, HBM / rocprofv3
section; both counters are in KiB.)  Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_...csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

# bench.py roofline site -> substring that identifies the kernel symbol (template arguments included)
SITES = {
    "edge_mlp_gemm1_fwd": "gemm_kernel<3, 2, 0, 0, 1, 1>",     # 48-row tiles, LN epilogue (M ~ 9000, N 256, K 384)
    "ffn_fwd_transformer_self": "ffn_fwd_kernel<false>",
    "scatter_add_fwd": "segment_reduce_kernel",
    "attention_fwd_transformer_self": "attn_fwd_stream_kernel<4>",
    "attention_fwd_transformer": "attn_fwd_stream_kernel<1>",
}


def per_kernel(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}, n


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 20 "
                "--warmup 3 --no-cpu-baseline --launch eager` (phonon_h128_b64), averaged per launch of the named "
                "kernel symbol; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, "
                "MI355X_MICROARCH.md HBM section). Kernel-symbol granularity: all launches of that symbol.",
       "sites": {}}
for site, key in SITES.items():
    ks = [k for k in fetch if key in k]
    if not ks:
        continue
    k = ks[0]
    f, w = fetch[k], write.get(k, 0.0)
    out["sites"][site] = {"kernel": key, "launches_profiled": nf[k], "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                          "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["sites"], indent=1))
