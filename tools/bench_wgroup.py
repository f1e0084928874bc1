#!/usr/bin/env python3
"""The weight-gradient flush groups of a cfg2 training step (Phonon L3 T2 H128 B64), launched ALONE through dosx_grad_flush
(finished mode: in-launch reduction over the M-splits): time per launch and fraction of the fp32 MFMA peak."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops  # noqa: E402

DEV = "cuda"
H, E, N, R2 = 128, 9344, 456, 6528


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


keep = []


def job(M, Nn, K, segs=None, **kw):
    dy = torch.randn(M, Nn, device=DEV)
    if segs is None:
        a = torch.randn(M, K, device=DEV)
        segs = [ops.seg(a)]
        keep.append(a)
    ns = ops.wgrad_splits(M, Nn, K)
    if M > 4096 and os.environ.get("WG_NS_E"):          # experiment: split count of the edge-row jobs
        ns = int(os.environ["WG_NS_E"])
    if M <= 4096 and os.environ.get("WG_NS_N"):         # experiment: split count of the node-row jobs
        ns = int(os.environ["WG_NS_N"])
    nsc = ops.wgrad_scratch_floats(Nn, K, ns)
    slab = torch.empty(max(nsc, 1), device=DEV)
    sb = torch.empty(ns * ((Nn + 63) // 64) * 64, device=DEV)
    dw, db = torch.empty(Nn, K, device=DEV), torch.empty(Nn, device=DEV)
    keep.extend([dy, slab, sb, dw, db, segs])
    return ops.wgrad_desc(M, Nn, ops.seg(dy), segs, slab, sb, ns, dst=dw, dst_bias=db, **kw)


def gnn_layer():
    x = torch.randn(N, H, device=DEV)
    e = torch.randn(E, H, device=DEV)
    src = torch.randint(0, N, (E,), device=DEV, dtype=torch.int32)
    dst = torch.sort(torch.randint(0, N, (E,), device=DEV, dtype=torch.int32))[0]
    keep.extend([x, e, src, dst])
    segs = [ops.seg(x, rmap=ops.rowmap(idx=src)), ops.seg(x, rmap=ops.rowmap(idx=dst)), ops.seg(e)]
    lnp = lambda K: dict(pro=ops.PRO_LN_PRELU, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV),
                         pro_alpha=torch.tensor([0.25], device=DEV))
    return [job(E, 2 * H, 3 * H, segs=segs), job(E, H, 2 * H, **lnp(2 * H)), job(N, 2 * H, 2 * H), job(N, H, 2 * H, **lnp(2 * H))]


def gnn_layer_factored():
    """the jobs of a message-passing layer with the first EdgeModel Linear factored (round 5: the shipped form): second Linear on
    the message gradient, the three column blocks of dW1 - source / destination node sums (N rows), dz on the edge rows -, the
    two NodeModel jobs"""
    lnp = lambda K: dict(pro=ops.PRO_LN_PRELU, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV),
                         pro_alpha=torch.tensor([0.25], device=DEV))
    return [job(E, H, 2 * H, **lnp(2 * H)), job(E, 2 * H, H), job(N, 2 * H, H), job(N, 2 * H, H), job(N, 2 * H, 2 * H),
            job(N, H, 2 * H, **lnp(2 * H))]


def enc_layer(R):
    rl = dict(pro=ops.PRO_ROWLN, pro_gamma=torch.randn(H, device=DEV), pro_beta=torch.randn(H, device=DEV),
              pro_stats=torch.rand(R, 2, device=DEV))
    return [job(R, 4 * H, H, **rl), job(R, H, 4 * H)]


def run(name, descs):
    descs = sorted(descs, key=lambda g: -(g.M * g.N * g.K))
    fl = sum(2.0 * d.M * d.N * d.K for d in descs)
    us = timeit(lambda: ops.grad_flush(descs, ()))
    wgs = sum(ops.wgrad_tiles(d.N, d.K) * d.nsplit for d in descs)
    print(f"wgroup {name:34s} jobs={len(descs):2d} workgroups={wgs:5d} {fl / 1e9:6.2f} GF: {us:7.1f} us  {fl / us / 1e6:6.1f} TF/s "
          f"({100 * fl / us / 1e6 / 157.3:4.1f}% of fp32 MFMA peak)")


if __name__ == "__main__":
    print("DOSX_WGRAD_OCC =", os.environ.get("DOSX_WGRAD_OCC", "(default)"), " DOSX_WGRAD_MAXSPLIT =", os.environ.get("DOSX_WGRAD_MAXSPLIT", "(default)"))
    run("GNN layer pair (factored)", gnn_layer_factored() + gnn_layer_factored())
    run("one GNN layer (factored)", gnn_layer_factored())
    run("GNN layer pair (gathered concat)", gnn_layer() + gnn_layer())
    run("one GNN layer (gathered concat)", gnn_layer())
    run("encoder stack T=2, 2B rows", enc_layer(R2) + enc_layer(R2))
    run("encoder stack T=2, B rows", enc_layer(R2 // 2) + enc_layer(R2 // 2))
    run("edge W1 alone", gnn_layer()[:1])
    run("fc1 alone", enc_layer(R2)[:1])
