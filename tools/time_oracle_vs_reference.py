#!/usr/bin/env python3
"""Side-by-side CPU timing of the ORACLE (oracle/dos_oracle.py, what `bench.py`'s `cpu_baseline` times on the GPU box) and
the IMPORTED REFERENCE (/root/reference, unmodified model files + the documented stand-ins of tests/golden/make_golden.py)
on the same batch, same weights, same thread count — BASELINE.md §2 / SURVEY.md §8d: "time both here, report the ratio".

Runs only in the build container (the reference never travels).  Full training steps (forward + loss + backward +
AdamW), steady state after warm-up.  Output is committed as profiles/r02_oracle_vs_reference.txt.

    python tools/time_oracle_vs_reference.py [--threads 8] [--budget 10]
"""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_golden as MG                                   # noqa: E402  (install_standins + the reference path)
from oracle import dos_oracle as O                         # noqa: E402
from dostransformer_amd import synth                       # noqa: E402

CASES = [  # (label, kind, L, T, H, B, dtype)
    ("cfg1 phonon L3 T1 H64 B8 fp64", "phonon", 3, 1, 64, 8, torch.float64),
    ("cfg2 phonon L3 T2 H128 B64 fp64", "phonon", 3, 2, 128, 64, torch.float64),
    ("cfg2 phonon L3 T2 H128 B64 fp32", "phonon", 3, 2, 128, 64, torch.float32),
    ("cfg3 eDOS L3 T2 H256 B16 fp32 (quarter batch)", "edos", 3, 2, 256, 16, torch.float32),
]


def timeit(fn, budget):
    fn()
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget or n >= 100:
            return el / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--budget", type=float, default=10.0)
    args = ap.parse_args()
    if not os.path.isdir(MG.REF):
        raise SystemExit(f"{MG.REF} not found: this script runs in the build container only")
    MG.install_standins()
    sys.path.insert(0, MG.REF)
    torch.set_num_threads(args.threads)
    print(f"# torch {torch.__version__}, {args.threads} threads, {os.cpu_count()} logical CPUs; full train steps (fwd+loss+bwd+AdamW)")
    print(f"{'case':52s} {'reference ms':>13s} {'oracle ms':>10s} {'oracle/reference':>17s}   loss(ref) loss(oracle)")
    for label, kind, L, T, H, B, dt in CASES:
        torch.set_default_dtype(dt)
        try:
            torch.manual_seed(0)
            if kind == "phonon":
                mod = importlib.import_module("embedder_phDOS.DOSTransformer_phonon")
                model = mod.DOSTransformer_phonon(L, T, 118, 4, H, "cpu", 0.0)
                g = synth.phonon_batch(B, seed=1000, dtype=dt)
            else:
                mod = importlib.import_module("embedder_eDOS.DOSTransformer")
                model = mod.DOSTransformer(L, T, 200, 41, 2, H, "cpu", 0.0)
                g = synth.edos_batch(B, seed=1000, dtype=dt)
        finally:
            torch.set_default_dtype(torch.float32)
        params = {k: v.detach().clone() for k, v in model.state_dict().items()}
        opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)
        last = {}

        def ref_step():
            # the reference's step body: main_phDOS.py:104-118 / main_eDOS.py:104-127
            model.train()
            opt.zero_grad()
            pg, _, ps = model(g)
            loss = O.loss_phonon(pg, ps, g.phdos, 1.0) if kind == "phonon" else O.loss_edos(pg, ps, g.y_ft, 1.0)
            loss.backward()
            opt.step()
            last["ref"] = float(loss)

        state = {}

        def oracle_step():
            l, _ = O.train_step(kind, params, state, g, L, T, lr=1e-4, beta=1.0)
            last["oracle"] = float(l)

        # first-step losses must agree (same weights, same batch) before any timing means anything
        ref_step()
        l_ref = last["ref"]
        oracle_step()
        l_or = last["oracle"]
        t_ref = timeit(ref_step, args.budget)
        t_or = timeit(oracle_step, args.budget)
        print(f"{label:52s} {1e3 * t_ref:13.1f} {1e3 * t_or:10.1f} {t_or / t_ref:17.3f}   {l_ref:.6f} {l_or:.6f}")


if __name__ == "__main__":
    main()
