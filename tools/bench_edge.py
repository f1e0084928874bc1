#!/usr/bin/env python3
"""The three launches of a factored message-passing layer (csrc/edge_mlp.hip) ALONE at a BASELINE shape: microseconds per launch
(HIP events around 200 back-to-back launches), next to the launches they replace; with the -DDOSX_STAMPS build
(DOSX_LIB=dostransformer_amd/csrc/build/libdosx_stamps.so) also the s_memtime phase stamps of one workgroup.
usage: bench_edge.py [phonon|edos] [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import _lib, functional as Fn, ops, synth  # noqa: E402
from dostransformer_amd.batch import bucket_sizes, collate, pad_batch  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "phonon"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = 128 if kind == "phonon" else int(os.environ.get("H", "128"))
dev = "cuda"
cs = synth.phonon_crystals(B, 0, torch.float32) if kind == "phonon" else synth.edos_crystals(B, 0, torch.float32)
g = collate(cs)
gp = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 8, 128)).to(dev)
m = gp.meta
N, E = m.num_nodes, m.num_edges
print(f"{kind} B={B}: N={N} E={E} H={H} tiles={m.seg_tile.shape[1] - 1}")
gen = torch.Generator().manual_seed(0)
P = {"k.0.weight": torch.randn(2 * H, 3 * H, generator=gen) * (3 * H) ** -0.5, "k.0.bias": torch.randn(2 * H, generator=gen),
     "k.1.weight": torch.randn(2 * H, generator=gen), "k.1.bias": torch.randn(2 * H, generator=gen),
     "k.2.weight": torch.tensor([0.25]), "k.3.weight": torch.randn(H, 2 * H, generator=gen) * (2 * H) ** -0.5,
     "k.3.bias": torch.randn(H, generator=gen)}
flat = torch.empty(sum((v.numel() + 3) // 4 * 4 for v in P.values()), device=dev)
o = 0
for k, v in list(P.items()):
    P[k] = flat[o:o + v.numel()].view(v.shape)
    P[k].copy_(v)
    o += (v.numel() + 3) // 4 * 4
x, e = torch.randn(N, H, device=dev), torch.randn(E, H, device=dev)
scale = m.inv_deg
W1, W3 = P["k.0.weight"], P["k.3.weight"]
pq = torch.empty(N, 4 * H, device=dev)
xhat, rstd = torch.empty(E, 2 * H, device=dev), torch.empty(E, device=dev)
agg, e_out = torch.empty(N, H, device=dev), torch.empty(E, H, device=dev)
dcat_n, de_next = torch.randn(N, 2 * H, device=dev), torch.randn(E, H, device=dev)
dmsg, dz, de = torch.empty(E, H, device=dev), torch.empty(E, 2 * H, device=dev), torch.empty(E, H, device=dev)
T = m.seg_tile.shape[1] - 1
part = torch.empty(T, 4 * H + 4, device=dev)
aggD, aggS, dx = torch.empty(N, 2 * H, device=dev), torch.empty(N, 2 * H, device=dev), torch.empty(N, H, device=dev)


def pair():
    ops.gemm_pair(dict(M=N, N=2 * H, segs=[ops.seg(x)], w=W1[:, :H], out=pq[:, :2 * H]),
                  dict(M=N, N=2 * H, segs=[ops.seg(x)], w=W1[:, H:2 * H], out=pq[:, 2 * H:]))


def fwd():
    ops.edge_mlp_fwd(E, H, e, pq, m.src, m.dst, W1[:, 2 * H:], P["k.0.bias"], P["k.1.weight"], P["k.1.bias"], P["k.2.weight"], W3,
                     P["k.3.bias"], xhat, rstd, e_out, m.seg_tile, m.rowptr_dst, scale, agg)


def fwd2():
    ops.gemm(E, 2 * H, [ops.seg(e)], W1[:, 2 * H:], xhat, bias=P["k.0.bias"], epi=ops.EPI_LN, aux_out=rstd, add_p=pq[:, :2 * H],
             add_ip=m.src, add_q=pq[:, 2 * H:], add_iq=m.dst)
    ops.gemm(E, H, [ops.seg(xhat)], W3, e_out, pro=ops.PRO_LN_PRELU, pro_gamma=P["k.1.weight"], pro_beta=P["k.1.bias"],
             pro_alpha=P["k.2.weight"], bias=P["k.3.bias"], res=e, epi=ops.EPI_SEGSUM, seg_tile=m.seg_tile, seg_rowptr=m.rowptr_dst,
             seg_scale=scale, seg_agg=agg)


def bwd():
    ops.edge_mlp_bwd(E, H, dcat_n[:, H:], de_next, m.dst, xhat, rstd, W3, W1[:, 2 * H:], P["k.1.weight"], P["k.1.bias"], P["k.2.weight"],
                     dmsg, dz, de, part, m.seg_tile, m.rowptr_dst, scale, aggD)


def bwd3():
    ops.edge_grad_combine(de_next, dcat_n.data_ptr() + 4 * H, 2 * H, m.dst, scale, dmsg, E, H)
    ops.gemm(E, 2 * H, [ops.seg(dmsg)], W3, dz, w_layout=1, epi=ops.EPI_PRELU_LN_BWD_SEG, aux=xhat, aux_stats=rstd,
             epi_gamma=P["k.1.weight"], epi_beta=P["k.1.bias"], epi_alpha=P["k.2.weight"], partials=part, partial_ld=4 * H + 4,
             seg_tile=m.seg_tile, seg_rowptr=m.rowptr_dst, seg_agg=aggD)
    ops.gemm(E, H, [ops.seg(dz)], W1[:, 2 * H:], de, w_layout=1, res=de_next)


def ngrad():
    ops.node_grad(N, H, dz, m.rowptr_src, m.perm_src, aggD, W1, dcat_n[:, :H], None, aggS, dx)


def ngrad2():
    ops.segment_reduce_perm(dz, m.rowptr_src, m.perm_src, aggS, N, E, 2 * H)
    ops.gemm(N, H, [ops.seg(aggS), ops.seg(aggD)], W1[:, :H], dx, w_layout=1, w_seg_off=H, res=dcat_n[:, :H])


def timeit(f, n=200):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


pair(); fwd(); bwd()
for name, f in (("gemm_pair (P | Q)", pair), ("edge_mlp_fwd", fwd), ("  = EPI_LN+add gemm, SEGSUM gemm", fwd2), ("edge_mlp_bwd", bwd),
                ("  = combine, PLB_SEG gemm, de gemm", bwd3), ("node_grad", ngrad), ("  = segment_sum_perm, w_seg_off gemm", ngrad2)):
    print(f"{name:40s} {timeit(f):7.2f} us")
lib = _lib.load()
if hasattr(lib, "dosx_debug_read_edge_stamps"):
    lib.dosx_debug_read_edge_stamps.argtypes = [C.c_void_p]
    fwd(); bwd(); ngrad()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 192)()
    lib.dosx_debug_read_edge_stamps(buf)
    for k, nm in enumerate(("edge_fwd", "edge_bwd", "node_grad")):
        s = [buf[k * 64 + i] for i in range(64)]
        t0 = s[0]
        print(nm, "matrix wave 0:", {i: int(s[i] - t0) for i in range(32) if s[i]})
        print(nm, "staging wave 0:", {i: int(s[32 + i] - t0) for i in range(32) if s[32 + i]})
