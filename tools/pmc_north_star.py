#!/usr/bin/env python3
"""The two counter-based figures BASELINE.json's north_star asks for, in one small file bench.py can put on its line:

  scatter_hbm_frac  achieved HBM rate of the scatter-add kernel / 8 TB/s, from the FETCH_SIZE / WRITE_SIZE passes over
                    `tools/bench_kernels.py --what scatter` (summarised by tools/pmc_scatter.py): the BASELINE configs[1]
                    size (9 000 edges, H 128) and the roofline-scale case (4 Mi edges, H 128, with the edge residual);
  attn_mfma_util    MFMA-busy cycles of all 1024 SIMDs / (1024 x the kernels' cycles at 2.4 GHz), from ONE pass with
                    SQ_VALU_MFMA_BUSY_CYCLES over `tools/bench_kernels.py --what attn`, forward + backward kernels of a
                    shape class together.  The classes are told apart by (template arguments, grid size): the key-tile
                    count NJ = 1 kernels are the 12-key Phonon-DOS cross attention, NJ = 4 at 2 query tiles x 128 entries the
                    51-key Phonon-DOS self attention, NJ = 3 / 4 at 7 query tiles the 41-key Electron-DOS cross attention,
                    NJ = 13 at 7 x 128 the 201-key Electron-DOS self attention (the roofline-scale launches are left out).
                    `*_aligned`: the same shape classes on the crystal-aligned kernels of csrc/attention_aligned.hip (round 5:
                    what dosx_attention_fwd / bwd run for <= 64 keys; the unsuffixed classes are attention.hip's kernels,
                    which the microbenchmark still times through dosx_attention_aligned_mode(0)).

The file carries the hash of the sources it was measured on (dostransformer_amd._lib.source_hash); bench.py reports the
figures only while that hash is the hash of the sources it runs - like roofline.traffic.
usage: pmc_north_star.py <attn counter_collection.csv> <scatter summary csv of pmc_scatter.py> <out.json> [git_head]"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS, GHZ = 256 * 4, 2.4


def attn_classes(path):
    rows = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = rows[r["Dispatch_Id"]]
        d["name"], d["grid"] = r["Kernel_Name"], int(r["Grid_Size"])
        d["lds"] = int(r.get("LDS_Block_Size", 0) or 0)
        d["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    agg = defaultdict(lambda: [0.0, 0.0, 0])
    for d in rows.values():
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
            continue
        # round 5: the crystal-aligned kernels (attention_aligned.hip; bench_kernels.py runs both forms).  NG = hidden / 64:
        # 4 = the Electron-DOS cross attention (41 / 64 keys), 2 = Phonon-DOS
        m = re.search(r"attn_al_(fwd|bwd)_kernel<(\d+)", d["name"])
        if m:
            ng, lds, wgs = int(m.group(2)), d.get("lds", 0), d["grid"] // 512
            if wgs > 512:
                continue
            if ng == 4:
                cls = "edos_cross_aligned"
            elif ng == 2:
                # (rocprofv3 reports 0 for dynamic LDS: the 51-key self attention is told from the 12-key cross attention of the
                #  same grid - 128 crystals x 2 workgroups - by its MFMA work, 4 x the key tiles)
                busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]
                self_ = wgs == 256 and busy > (2.0e6 if m.group(1) == "fwd" else 5.0e6)
                cls = "cfg2_self_aligned" if self_ else "cfg2_cross_aligned"
            else:
                continue
            a = agg[cls]
            a[0] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
            a[1] += d["ns"]
            a[2] += 1
            continue
        m = re.search(r"attn_(fwd_stream|bwd_dq_stream|bwd_dkv)_kernel<(\d+)", d["name"])
        if not m:
            continue
        nj, wgs = int(m.group(2)), d["grid"] // 512
        if m.group(1) == "bwd_dkv":
            cls = "edos_self" if nj == 2 and wgs <= 4 * 128 else None      # KG = 2 key groups x 128 crystals; roofline scale: more
        elif nj == 1:
            cls = "cfg2_cross"
        elif nj in (3, 4):           # (NJ = 3: the 33-48-key row phases of round 4 - the 41-key Electron-DOS cross attention)
            cls = "cfg2_self" if (nj == 4 and wgs <= 2 * 128) else ("edos_cross" if 2 * 128 < wgs <= 7 * 128 else None)
        elif nj == 13:
            cls = "edos_self" if wgs <= 7 * 128 else None
        else:
            cls = None
        if cls is None:
            continue
        a = agg[cls]
        a[0] += d["SQ_VALU_MFMA_BUSY_CYCLES"]
        a[1] += d["ns"]
        a[2] += 1
    # structural check of the work-threshold classification above: tools/bench_kernels.py --what attn runs two 12-key
    # cross cases, one 51-key self case and three Electron-DOS cross cases, each the same number of aligned launches - a launch filed
    # under the wrong class breaks the 2 : 1 : 3 ratio
    n_self = agg["cfg2_self_aligned"][2] if "cfg2_self_aligned" in agg else 0
    if n_self:
        got = (agg["cfg2_cross_aligned"][2], n_self, agg["edos_cross_aligned"][2])
        if got != (2 * n_self, n_self, 3 * n_self):
            raise SystemExit(f"pmc_north_star: aligned attention launches per class {got}, expected the ratio 2 : 1 : 3 - "
                             "the self / cross classification thresholds no longer fit the kernels")
    return {k: {"util": round(b / (SIMDS * ns * GHZ), 4), "launches": n} for k, (b, ns, n) in sorted(agg.items())}


def scatter_classes(path):
    out = {}
    for r in csv.DictReader(open(path)):
        grid, frac = int(r["grid_size"]), round(float(r["pct_of_8TBs"]) / 100.0, 4)
        mb = float(r["HBM_MB_per_launch"])
        if grid < 100000:
            out["cfg2"] = {"frac": frac, "avg_us": float(r["avg_us"]), "hbm_mb": mb}
        elif "4Mi" not in out and 6000 < mb < 7000 and grid < 10_000_000:     # 4 Mi edges x H 128 with the edge residual: 6.55 GB
            out["4Mi"] = {"frac": frac, "avg_us": float(r["avg_us"]), "hbm_mb": mb}
    return out


def main():
    from dostransformer_amd._lib import source_hash
    att, sc = attn_classes(sys.argv[1]), scatter_classes(sys.argv[2])
    out = {"_note": __doc__.split("usage:")[0].strip(), "source_hash": source_hash(),
           "git_head": sys.argv[4] if len(sys.argv) > 4 else "unknown",
           "scatter_hbm_frac": {k: v["frac"] for k, v in sc.items()}, "scatter_detail": sc,
           "attn_mfma_util": {k: v["util"] for k, v in att.items()}, "attn_detail": att}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("scatter_hbm_frac", "attn_mfma_util", "source_hash")}))


if __name__ == "__main__":
    main()
