#!/usr/bin/env python3
"""Can the vector ALU carry GEMM work WHILE MFMA kernels of the chain run on the same CUs?  Stream A: a loop of large MFMA
GEMMs (dosx_gemm, the Electron-DOS feed-forward shapes: one 8-wave workgroup per CU or two, 117-250 VGPRs); stream B: a
loop of vector-ALU sliver GEMMs (4 waves, 56 VGPRs, 8.5 KB of LDS).  Each alone, then both together: time of each stream and
the aggregate rate."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dostransformer_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def main():
    lib = _lib.load()
    A, B = torch.cuda.Stream(), torch.cuda.Stream()
    M = 25728
    cases = {
        "fc1 fwd  N1024 K256 (ROWLN prologue)": dict(N=1024, K=256, wl=0, pro=ops.PRO_ROWLN),
        "fc2 fwd  N256 K1024": dict(N=256, K=1024, wl=0, pro=0),
        "fc1 dgrad N256 K1024 (w_layout 1)": dict(N=256, K=1024, wl=1, pro=0),
    }
    # sliver work: plain dgrad GEMMs (w_layout 1), 1.69 GF each
    Ms, Ns, Ks = 12864, 256, 256
    sa, sw, sc = torch.randn(Ms, Ks, device=DEV), torch.randn(Ks, Ns, device=DEV), torch.empty(Ms, Ns, device=DEV)
    n_s = 40

    def sliver_loop():
        lib.dosx_set_sliver_max_gf(2.0)
        for _ in range(n_s):
            ops.gemm(Ms, Ns, [ops.seg(sa)], sw, sc, w_layout=1)
        lib.dosx_set_sliver_max_gf(0.0)

    def timed(fa, fb):
        torch.cuda.synchronize()
        ea0, ea1, eb0, eb1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        if fa is not None:
            with torch.cuda.stream(A):
                ea0.record(); fa(); ea1.record()
        if fb is not None:
            with torch.cuda.stream(B):
                eb0.record(); fb(); eb1.record()
        torch.cuda.synchronize()
        return (ea0.elapsed_time(ea1) * 1e3 if fa is not None else 0.0), (eb0.elapsed_time(eb1) * 1e3 if fb is not None else 0.0)

    for _ in range(2):
        timed(None, sliver_loop)
    _, tb = timed(None, sliver_loop)
    fls = 2.0 * Ms * Ns * Ks * n_s
    print(f"sliver GEMMs alone: {n_s} x 1.69 GF in {tb:.0f} us = {fls / tb / 1e6:.1f} TF/s")
    # the co-residable MFMA kernel: a weight-gradient group (2 workgroups of 107 VGPRs / 75 KB per CU leave room for a sliver)
    keep = []

    def job(Mw, Nw, Kw):
        dy, a_ = torch.randn(Mw, Nw, device=DEV), torch.randn(Mw, Kw, device=DEV)
        ns = ops.wgrad_splits(Mw, Nw, Kw)
        slab = torch.empty(max(ops.wgrad_scratch_floats(Nw, Kw, ns), 1), device=DEV)
        sb = torch.empty(ns * ((Nw + 63) // 64) * 64, device=DEV)
        dw, db = torch.empty(Nw, Kw, device=DEV), torch.empty(Nw, device=DEV)
        keep.extend([dy, a_, slab, sb, dw, db])
        return ops.wgrad_desc(Mw, Nw, ops.seg(dy), [ops.seg(a_)], slab, sb, ns, dst=dw, dst_bias=db)
    group = [job(M, 1024, 256), job(M, 256, 1024), job(M, 1024, 256), job(M, 256, 1024)]
    n_g = 4

    def wgrad_loop():
        for _ in range(n_g):
            ops.wgrad_grouped(group)
    for _ in range(2):
        timed(wgrad_loop, None)
    tg, _ = timed(wgrad_loop, None)
    flg = 2.0 * M * 1024 * 256 * 4 * n_g
    tg2, tb2 = timed(wgrad_loop, sliver_loop)
    tot = max(tg2, tb2)
    print(f"weight-gradient groups: alone {tg:.0f} us ({flg / tg / 1e6:.1f} TF/s) | together: groups {tg2:.0f} us, sliver {tb2:.0f} us -> "
          f"aggregate {(flg + fls) / tot / 1e6:.1f} TF/s in {tot:.0f} us (serial would be {tg + tb:.0f} us)")
    for name, c in cases.items():
        N, K = c["N"], c["K"]
        a = torch.randn(M, K, device=DEV)
        w = torch.randn(N, K, device=DEV) if c["wl"] == 0 else torch.randn(K, N, device=DEV)
        out = torch.empty(M, N, device=DEV)
        kw = {}
        if c["pro"] == ops.PRO_ROWLN:
            kw = dict(pro=ops.PRO_ROWLN, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV),
                      pro_stats=torch.rand(M, 2, device=DEV))
        n_a = 12

        def mfma_loop():
            for _ in range(n_a):
                ops.gemm(M, N, [ops.seg(a)], w, out, w_layout=c["wl"], **kw)
        for _ in range(2):
            timed(mfma_loop, None)
        ta, _ = timed(mfma_loop, None)
        fla = 2.0 * M * N * K * n_a
        ta2, tb2 = timed(mfma_loop, sliver_loop)
        tot = max(ta2, tb2)
        print(f"{name}: MFMA alone {ta:.0f} us ({fla / ta / 1e6:.1f} TF/s) | together: MFMA {ta2:.0f} us, sliver {tb2:.0f} us -> "
              f"aggregate {(fla + fls) / tot / 1e6:.1f} TF/s in {tot:.0f} us (serial would be {ta + tb:.0f} us)")


if __name__ == "__main__":
    main()
