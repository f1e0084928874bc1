#!/usr/bin/env python3
"""north_star figures of the kernels that RUN in the replayed training steps: MFMA utilisation of the
launches that contain the attention, achieved bytes / time of the launch that contains the scatter-add.

Inputs: ONE rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over `python3 bench.py ... --no-secondary` per
configuration (the replayed step itself, program directly after `--`), and the per-symbol HBM bytes of the FETCH_SIZE / WRITE_SIZE
passes over the same command (tools/pmc_traffic.py's json).

  mfma_util   MFMA-busy cycles summed over the chip's 1024 SIMDs / (1024 x the kernel's own cycles at the nominal 2.4 GHz), per
              kernel symbol, launch-weighted - tools/pmc_mfma.py's definition; *_active: / GRBM_GUI_ACTIVE per XCD instead
  hbm         (2 FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction of MI355X_MICROARCH.md) / the symbol's average
              duration in the counter pass / 8 TB/s; `algorithmic`: the launch's algorithmic bytes (E H w read of e,
              E 2H w xhat, E H w e', the node-row gathers, N H w aggregate) / the same duration

usage: pmc_step.py <out.json> <git_head> <cfg2 mfma csv> <cfg2 traffic json> [<edos mfma csv> <edos traffic json>]"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS, GHZ, XCDS, HBM = 256 * 4, 2.4, 8, 8.0e12


def short(sym):
    s = sym.replace("(anonymous namespace)::", "").replace("void ", "")
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i].strip()
    return s.strip()


def per_symbol(path):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = rows[r["Dispatch_Id"]]
        d["name"] = short(r["Kernel_Name"])
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in rows.values():
        a = agg[d["name"]]
        a[0] += 1
        a[1] += d["ns"]
        a[2] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[3] += d.get("GRBM_GUI_ACTIVE", 0.0)
    out = {}
    for k, (n, ns, busy, gui) in agg.items():
        out[k] = {"launches": n, "avg_us": round(ns / n / 1e3, 2), "mfma_util": round(busy / (SIMDS * ns * GHZ), 4) if ns else 0.0,
                  "mfma_util_active": round(busy / (SIMDS * gui / XCDS), 4) if gui > 0 else None}
    return out


def pick(tab, pattern):
    """launch-weighted figures over the symbols that match `pattern`"""
    sel = {k: v for k, v in tab.items() if re.search(pattern, k)}
    n = sum(v["launches"] for v in sel.values())
    if not n:
        return None
    us = sum(v["avg_us"] * v["launches"] for v in sel.values())
    busy = sum(v["mfma_util"] * v["avg_us"] * v["launches"] for v in sel.values())
    return {"mfma_util": round(busy / us, 4), "avg_us": round(us / n, 2), "launches": n, "symbols": sorted(sel)}


def main():
    from dostransformer_amd._lib import source_hash
    out_path, head = sys.argv[1:3]
    cfgs = {"cfg2": sys.argv[3:5]}
    if len(sys.argv) >= 7:
        cfgs["cfg3"] = sys.argv[5:7]
    out = {"_note": __doc__.split("usage:")[0].strip(), "source_hash": source_hash(), "git_head": head, "per_symbol": {}, "in_step": {}}
    for name, (mfma_csv, traffic_json) in cfgs.items():
        tab = per_symbol(mfma_csv)
        out["per_symbol"][name] = {k: v for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]) if v["mfma_util"] > 0}
        st = {}
        # the launches that contain the attention (layers/multihead_attention.py:68-72)
        for key, pat in (("ffn_fwd_att", r"ffn_fwd(_multi)?_kernel<.*, [12]>$"), ("ffn_bwd_att", r"ffn_bwd_kernel<.*, 2>$"),
                         ("attn_al_fwd", r"attn_al_fwd_kernel"), ("attn_al_bwd", r"attn_al_bwd_kernel"),
                         ("attn_fwd_stream", r"attn_fwd_stream_kernel"), ("attn_bwd_dq_stream", r"attn_bwd_dq_stream_kernel"),
                         ("attn_bwd_dkv", r"attn_bwd_dkv_kernel"), ("edge_fwd", r"edge_fwd_kernel"), ("edge_bwd", r"edge_bwd_kernel"),
                         ("wgrad_grouped", r"wgrad_grouped_kernel")):
            p = pick(tab, pat)
            if p:
                st[key] = {"mfma_util": p["mfma_util"], "avg_us": p["avg_us"], "launches": p["launches"]}
        # the launch that contains the scatter-add (DOSTransformer_phonon.py:209): HBM bytes from the traffic passes / its duration
        try:
            tr = json.load(open(traffic_json)).get("kernels", {})
        except Exception:
            tr = {}
        for sym, rec in tr.items():
            if sym.startswith("edge_fwd_kernel") and sym in tab:
                us = tab[sym]["avg_us"]
                st["scatter_in_edge_fwd"] = {"kernel": sym, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "avg_us": us,
                                             "hbm_frac": round(rec["hbm_bytes_per_launch"] / (us * 1e-6) / HBM, 4)}
            if sym.startswith("segment_reduce_kernel") and sym in tab:
                us = tab[sym]["avg_us"]
                st["scatter_add"] = {"kernel": sym, "hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "avg_us": us,
                                     "hbm_frac": round(rec["hbm_bytes_per_launch"] / (us * 1e-6) / HBM, 4)}
        out["in_step"][name] = st
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out["in_step"]))


if __name__ == "__main__":
    main()
