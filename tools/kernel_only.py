#!/usr/bin/env python3
"""Kernel-only roofline fractions of a bench run's sites: the site's algorithmic work per step (bench.py's
per-site table) / the summed durations of its kernel symbol per step in the rocprofv3 --kernel-trace --stats run of the same
command - no event brackets, no dispatch gaps, no waiting for CUs under another stream's kernels.  Steps of the traced run = calls
of adamw_kernel (one per step).  Sites that share a kernel symbol (dosx_gemm's template instantiations are one symbol each; the
ffn kernels have one symbol per tile form) are matched by the symbol prefix the site names, share ONE fraction (their summed work
over the symbol's summed time) and say so in `shares_symbol_with`; a gemm site also owns the mixed-tile-height instantiation
of its template arguments (gemm_mixed_kernel).

usage: kernel_only.py <kernel_stats.csv> <sites.json> <out.json> [git_head]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = {"mfma": 157.3e12, "hbm": 8.0e12}


def main():
    from dostransformer_amd._lib import source_hash
    stats, sites_path, out_path = sys.argv[1:4]
    head = sys.argv[4] if len(sys.argv) > 4 else "unknown"
    rows = []
    for r in csv.DictReader(open(stats)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        rows.append((name, int(r["Calls"]), float(r["TotalDurationNs"])))
    steps = next((c for n, c, _ in rows if n.startswith("adamw_kernel")), None)
    if not steps:
        raise SystemExit("no adamw_kernel row: cannot tell the number of steps")
    sites = json.load(open(sites_path))
    by_kernel = {}
    for s in sites["sites"]:
        by_kernel.setdefault(s["kernel"], []).append(s)
    out = {"_note": __doc__.split("usage:")[0].strip(), "source_hash": source_hash(), "git_head": head, "config": sites.get("config"),
           "steps_traced": steps, "sites": {}}
    def runs_as(kernel, name):
        """Whether the traced symbol ``name`` is a launch of the site kernel ``kernel``.  dosx_gemm runs a problem whose last round
        of workgroups is partial as gemm_mixed_kernel<RT, RT_tail, ...rest> - the site table names the plain instantiation
        gemm_kernel<RT, ...rest> (ops.gemm cannot see the library's choice)."""
        if name.startswith(kernel):
            return True
        m = re.match(r"gemm_kernel<(\d+), (.*)>$", kernel)
        return bool(m and re.match(r"gemm_mixed_kernel<%s, \d+, %s>" % (m.group(1), re.escape(m.group(2))), name))

    for kernel, group in by_kernel.items():
        sel = [(n, c, t) for n, c, t in rows if runs_as(kernel, n)]
        if not sel:
            continue
        calls, total = sum(c for _, c, _ in sel), sum(t for _, _, t in sel)
        work = sum(s["work_per_launch"] * s.get("brackets_per_step", s["launches_per_step"]) for s in group)     # per step
        us_step = total / steps / 1e3
        frac = work / (us_step * 1e-6) / PEAK[group[0]["bound"]] if us_step > 0 else None
        for s in group:
            out["sites"][s["site"]] = {"kernel": kernel, "kernels_per_step": round(calls / steps, 2), "avg_kernel_us": round(total / calls / 1e3, 2),
                                       "us_per_step": round(us_step, 2), "frac": round(frac, 4), "shares_symbol_with": len(group) - 1}
    json.dump(out, open(out_path, "w"), indent=1)
    dom = max(out["sites"].items(), key=lambda kv: kv[1]["us_per_step"])
    print("steps", steps, "| largest:", dom[0], dom[1])


if __name__ == "__main__":
    main()
