#!/usr/bin/env python3
"""Micro-benchmarks of the libdosx kernels at the shapes of the BASELINE configs (and at a
roofline scale that leaves the 256 MiB Infinity Cache).  Prints one line per case:
achieved TFLOP/s (fp32 MFMA peak 157.3) or GB/s (HBM peak 8000)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dostransformer_amd import ops  # noqa: E402

DEV = "cuda"


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
    return s.elapsed_time(e) * 1e3 / iters     # us


def bf16x3_case(name, M, N, K, wl):
    a = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) if wl == 0 else torch.randn(K, N, device=DEV)
    out, out32 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    us3 = timeit(lambda: ops.gemm_bf16x3(a, w, out, w_layout=wl))
    us32 = timeit(lambda: ops.gemm(M, N, [ops.seg(a)], w, out32, w_layout=wl))
    rows = slice(0, min(M, 4096))
    ref = a[rows].double() @ (w.double().T if wl == 0 else w.double())
    e3 = float((out[rows].double() - ref).abs().max() / ref.abs().max())
    e32 = float((out32[rows].double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * K
    print(f"bf16x3 {name:18s} M={M:6d} N={N:5d} K={K:5d} wl={wl}: split-bf16 {us3:8.1f} us {fl / us3 / 1e6:6.1f} TF/s(fp32-equivalent) | "
          f"fp32 MFMA {us32:8.1f} us {fl / us32 / 1e6:6.1f} TF/s | x{us32 / us3:4.2f} | max err vs fp64: {e3:.2e} / {e32:.2e}", flush=True)


def gemm_case(name, M, N, K, wl=0, pro=0, epi=0, nseg=1, gather=False):
    a = torch.randn(M, K // nseg, device=DEV)
    segs = [ops.seg(a)] * nseg
    if gather:
        n_nodes = max(1, M // 20)
        x = torch.randn(n_nodes, K // 3, device=DEV)
        src = torch.randint(0, n_nodes, (M,), device=DEV, dtype=torch.int32)
        dst = torch.sort(torch.randint(0, n_nodes, (M,), device=DEV, dtype=torch.int32))[0]
        e = torch.randn(M, K // 3, device=DEV)
        segs = [ops.seg(x, rmap=ops.rowmap(idx=src)), ops.seg(x, rmap=ops.rowmap(idx=dst)), ops.seg(e)]
    w = torch.randn(N, K, device=DEV) if wl == 0 else torch.randn(K, N, device=DEV)
    out = torch.empty(M, N, device=DEV)
    kw = {}
    if epi == ops.EPI_LN:
        kw = dict(epi=ops.EPI_LN, aux_out=torch.empty(M, device=DEV), bias=torch.randn(N, device=DEV))
    if epi == ops.EPI_PRELU_LN_BWD:
        rows = ops.gemm_partial_rows(M, N, epi)
        kw = dict(epi=epi, aux=torch.randn(M, N, device=DEV), aux_stats=torch.rand(M, device=DEV) + 0.5,
                  epi_gamma=torch.randn(N, device=DEV), epi_beta=torch.randn(N, device=DEV),
                  epi_alpha=torch.tensor([0.25], device=DEV), partials=torch.empty(rows, 2 * N + 4, device=DEV),
                  partial_ld=2 * N + 4)
    if epi == ops.EPI_PRELU_BWD:
        rows = ops.gemm_partial_rows(M, N, epi)
        kw = dict(epi=epi, aux=torch.randn(M, N, device=DEV), epi_alpha=torch.tensor([0.25], device=DEV),
                  partials=torch.empty(rows, 4, device=DEV), partial_ld=4)
    if pro == ops.PRO_LN_PRELU:
        kw.update(pro=pro, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV),
                  pro_alpha=torch.tensor([0.25], device=DEV))
    if pro == ops.PRO_ROWLN:
        kw.update(pro=pro, pro_gamma=torch.randn(K, device=DEV), pro_beta=torch.randn(K, device=DEV),
                  pro_stats=torch.rand(M, 2, device=DEV))
    us = timeit(lambda: ops.gemm(M, N, segs, w, out, w_layout=wl, **kw))
    tf = 2.0 * M * N * K / us / 1e6
    print(f"gemm  {name:34s} M={M:6d} N={N:4d} K={K:4d} wl={wl} pro={pro} epi={epi}: {us:8.1f} us  {tf:6.1f} TF/s  "
          f"({100 * tf / 157.3:4.1f}% of fp32 MFMA peak)")


def wgrad_case(name, M, N, K):
    dy = torch.randn(M, N, device=DEV)
    a = torch.randn(M, K, device=DEV)
    ns = ops.wgrad_splits(M, N, K)
    # finished mode, as in the training step: the kernel reduces its M-splits itself and writes dW / db (the time includes it)
    nf = ops.wgrad_scratch_floats(N, K, ns)
    slab = torch.empty(max(nf, 1), device=DEV)
    sb = torch.empty(ns * ((N + 63) // 64) * 64, device=DEV)
    dw, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    g = ops.wgrad_desc(M, N, ops.seg(dy), [ops.seg(a)], slab, sb, ns, dst=dw, dst_bias=db)
    us = timeit(lambda: ops.wgrad_grouped([g]))
    tf = 2.0 * M * N * K / us / 1e6
    print(f"wgrad {name:34s} M={M:6d} N={N:4d} K={K:4d} splits={ns:2d}: {us:8.1f} us  {tf:6.1f} TF/s  "
          f"({100 * tf / 157.3:4.1f}%)")


def segreduce_case(name, N, deg, H, residual=True):
    E = N * deg
    msg = torch.randn(E, H, device=DEV)
    e_in = torch.randn(E, H, device=DEV) if residual else None
    e_out = torch.empty(E, H, device=DEV) if residual else None
    rowptr = (torch.arange(N + 1, device=DEV, dtype=torch.int32) * deg).contiguous()
    agg = torch.empty(N, H, device=DEV)
    us = timeit(lambda: ops.segment_reduce(msg, rowptr, None, agg, e_in, e_out, N, E, H))
    by = 4.0 * (E * H + N + 1 + N * H + (2 * E * H if residual else 0))
    print(f"scatter-add {name:28s} N={N:8d} E={E:9d} H={H} residual={int(residual)}: {us:9.1f} us  "
          f"{by / us / 1e3:7.1f} GB/s ({100 * by / us / 1e3 / 8000:4.1f}% of 8 TB/s)  [{by / 1e6:.1f} MB]")


def attn_case(name, Sq, Bq, Nk, Bk, H):
    from dostransformer_amd._lib import Attn
    x = torch.randn(Sq * Bq, H, device=DEV)
    kv = torch.randn(Nk * Bk, H, device=DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    out = torch.empty(Sq * Bq, H, device=DEV)
    probs = torch.empty(Bq, Sq, Nk, device=DEV)
    qs, os_ = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, Bq, 1
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), g.data_ptr(), b.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qs.data_ptr(), os_.data_ptr()
    us = timeit(lambda: ops.attention_fwd(a))
    fl = 4.0 * Bq * Sq * Nk * H
    dout = torch.randn(Sq * Bq, H, device=DEV)
    dx = torch.empty(Sq * Bq, H, device=DEV)
    dsc = torch.empty(Bq, Sq, Nk, device=DEV)
    dkv = torch.zeros(Nk * Bk, H, device=DEV)
    nqt, nkt = (Sq + 31) // 32, (Nk + 31) // 32
    part = torch.empty(Bq * nqt + Bk * max(nkt, (Nk + 15) // 16), 2 * H, device=DEV)
    a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), dsc.data_ptr(), dkv.data_ptr(), 1
    a.partials_q, a.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * H
    usb = timeit(lambda: ops.attention_bwd(a))
    note = ""
    from dostransformer_amd import _lib
    if _lib.load().dosx_attention_pkv_supported(Nk, H):      # the path the training programs take for Nk <= 64
        kvp = torch.empty(Bq * nqt * Nk, H, device=DEV)
        a.dkv_part = kvp.data_ptr()
        usp = timeit(lambda: ops.attention_bwd(a))
        note = f" | bwd(dq + in-kernel dK/dV partials + reduce) {usp:7.1f} us {2.5 * fl / usp / 1e6:6.2f} TF/s"
        # the ONE-launch form of the training step (counters), and the crystal-aligned kernels behind the same two entry points
        # (csrc/attention_aligned.hip; mode 2 = every shape they take, 0 = attention.hip's kernels)
        lib = _lib.load()
        a.dkv_cnt = ops.COUNTERS.take(DEV, Bk)
        res = {}
        for mode in (0, 2):
            prev = lib.dosx_attention_aligned_mode(mode)
            res[mode] = (timeit(lambda: ops.attention_fwd(a)), timeit(lambda: ops.attention_bwd(a)))
            lib.dosx_attention_aligned_mode(prev)
        note += (f" || one launch: fwd {res[0][0]:6.1f} bwd {res[0][1]:6.1f} us | crystal-aligned tiles: fwd {res[2][0]:6.1f} us "
                 f"({100 * fl / res[2][0] / 1e6 / 157.3:4.1f}%) bwd {res[2][1]:6.1f} us ({100 * 2.5 * fl / res[2][1] / 1e6 / 157.3:4.1f}%)")
    print(f"attn  {name:30s} Sq={Sq} Bq={Bq} Nk={Nk} H={H}: fwd {us:7.1f} us {fl / us / 1e6:6.2f} TF/s "
          f"({100 * fl / us / 1e6 / 157.3:4.1f}%) | bwd(dq+streamed dkv) {usb:7.1f} us {2.5 * fl / usb / 1e6:6.2f} TF/s" + note)


def ffn_case(name, M, H):
    x = torch.randn(M, H, device=DEV); stats = torch.rand(M, 2, device=DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    w1, b1 = torch.randn(4 * H, H, device=DEV), torch.randn(4 * H, device=DEV)
    w2, b2 = torch.randn(H, 4 * H, device=DEV), torch.randn(H, device=DEV)
    h = torch.empty(M, 4 * H, device=DEV); out = torch.empty(M, H, device=DEV)
    us = timeit(lambda: ops.ffn_fwd(M, H, x, stats, g, b, w1, b1, w2, b2, h, out))
    def two():
        ops.gemm(M, 4 * H, [ops.seg(x)], w1, h, pro=ops.PRO_ROWLN, pro_gamma=g, pro_beta=b, pro_stats=stats, bias=b1, act=1)
        ops.gemm(M, H, [ops.seg(h)], w2, out, bias=b2, res=x)
    us2 = timeit(two)
    fl = 16.0 * M * H * H
    print(f"ffn   {name:30s} M={M} H={H}: fused {us:6.1f} us {fl / us / 1e6:6.1f} TF/s ({100 * fl / us / 1e6 / 157.3:4.1f}%) | "
          f"two GEMMs {us2:6.1f} us")


def layer_case(name, Sq, Bq, Nk, Bk, H):
    """one cross-attention encoder layer forward: dosx_attention_fwd + dosx_ffn_fwd against the one-launch form (DosxFfn.att_*)"""
    from dostransformer_amd._lib import Attn
    M = Sq * Bq
    x = torch.randn(M, H, device=DEV)
    kv = torch.randn(Nk * Bk, H, device=DEV)
    g0, b0, g1, b1n = (torch.randn(H, device=DEV) for _ in range(4))
    w1, b1 = torch.randn(4 * H, H, device=DEV), torch.randn(4 * H, device=DEV)
    w2, b2 = torch.randn(H, 4 * H, device=DEV), torch.randn(H, device=DEV)
    h = torch.empty(M, 4 * H, device=DEV); out = torch.empty(M, H, device=DEV); x1 = torch.empty(M, H, device=DEV)
    probs = torch.empty(Bq, Sq, Nk, device=DEV)
    qs, st1 = torch.empty(M, 2, device=DEV), torch.empty(M, 2, device=DEV)
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, Bq, 1
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), g0.data_ptr(), b0.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = x1.data_ptr(), probs.data_ptr(), qs.data_ptr(), st1.data_ptr()

    def two():
        ops.attention_fwd(a)
        ops.ffn_fwd(M, H, x1, st1, g1, b1n, w1, b1, w2, b2, h, out)
    att = dict(kvhat=kv, gamma0=g0, beta0=b0, Nk=Nk, Bk=Bk, Bq=Bq, Sq=Sq, qs=Bq, qb=1, probs=probs, qstats=qs, x1=x1, st1=st1)
    us2 = timeit(two)
    usa = timeit(lambda: ops.attention_fwd(a))
    usf = timeit(lambda: ops.ffn_fwd(M, H, x1, st1, g1, b1n, w1, b1, w2, b2, h, out))
    us1 = timeit(lambda: ops.ffn_fwd(M, H, x, None, g1, b1n, w1, b1, w2, b2, h, out, att=att)) if ops.ffn_att_supported(H, Nk) else float("nan")
    usl = float("nan")
    if ops.ffn_att_aligned_supported(H, Nk):
        att2 = dict(att, aligned=True)
        usl = timeit(lambda: ops.ffn_fwd(M, H, x, None, g1, b1n, w1, b1, w2, b2, h, out, att=att2))
    print(f"layer {name:30s} Sq={Sq} Bq={Bq} Nk={Nk} H={H}: one launch {us1:6.1f} us (per-row keys) / {usl:6.1f} us (crystal-aligned tiles) | "
          f"attention {usa:5.1f} + ffn {usf:5.1f} us, back to back {us2:6.1f} us")


def ffn_bwd_case(name, M, H):
    x = torch.randn(M, H, device=DEV); stats = torch.rand(M, 2, device=DEV)
    g = torch.randn(H, device=DEV)
    w1, w2 = torch.randn(4 * H, H, device=DEV), torch.randn(H, 4 * H, device=DEV)
    h = torch.randn(M, 4 * H, device=DEV); dy = torch.randn(M, H, device=DEV)
    dh = torch.empty(M, 4 * H, device=DEV); dx = torch.empty(M, H, device=DEV)
    part = torch.empty(ops.ffn_bwd_partial_rows(M), 2 * H, device=DEV)
    us = timeit(lambda: ops.ffn_bwd(M, H, dy, h, x, stats, g, w1, w2, dh, dx, part))
    part2 = torch.empty(ops.gemm_partial_rows(M, H, ops.EPI_ROWLN_BWD), 2 * H, device=DEV)
    def two():
        ops.gemm(M, 4 * H, [ops.seg(dy)], w2, dh, w_layout=1, epi=ops.EPI_RELU_MASK, aux=h)
        ops.gemm(M, H, [ops.seg(dh)], w1, dx, w_layout=1, epi=ops.EPI_ROWLN_BWD, aux=x, aux_stats=stats, epi_gamma=g, res=dy,
                 partials=part2, partial_ld=2 * H)
    us2 = timeit(two)
    fl = 16.0 * M * H * H
    print(f"ffnb  {name:30s} M={M} H={H}: fused {us:6.1f} us {fl / us / 1e6:6.1f} TF/s ({100 * fl / us / 1e6 / 157.3:4.1f}%) | "
          f"two GEMMs {us2:6.1f} us")


def csr_case(name, B, kind="phonon"):
    from dostransformer_amd import synth
    g = synth.phonon_batch(B, seed=0, dtype=torch.float32, sort_edges=False) if kind == "phonon" else \
        synth.edos_batch(B, seed=0, dtype=torch.float32, sort_edges=False)
    ei, bv = g.edge_index.to(DEV), g.batch.to(DEV)
    us = timeit(lambda: ops.csr_build(ei, bv, B), iters=20)
    print(f"csr   {name:30s} B={B} N={bv.shape[0]} E={ei.shape[1]}: {us:7.1f} us per build (2 stable radix sorts + 5 kernels, "
          f"no host round trip)")


def collate_case(name, B, kind="phonon"):
    import time
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.loader import DeviceDataset
    cs = synth.phonon_crystals(4 * B, seed=0, dtype=torch.float32) if kind == "phonon" else synth.edos_crystals(4 * B, seed=0, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    sel = list(range(B, 2 * B))
    for _ in range(3):
        ds.collate(sel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        ds.collate(sel)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(10):
        collate([cs[i] for i in sel]).to(DEV)
    torch.cuda.synchronize()
    t_ref = (time.perf_counter() - t0) / 10
    print(f"coll  {name:30s} B={B}: device collate {t_dev * 1e6:7.1f} us per batch (host part {t_host * 1e6:6.1f} us) | host collate + "
          f"H2D {t_ref * 1e6:8.1f} us")


def neighbor_case(name, C, r_max):
    """A phonon-dataset-sized featurisation (the reference's set is ~1.5k crystals, r_max 4, `main_phDOS.py:21`)."""
    import time
    import numpy as np
    rng = np.random.default_rng(0)
    sizes = rng.integers(2, 13, C)
    pos, cells = [], []
    for n in sizes:
        cell = np.diag(rng.uniform(3.0, 7.0, 3)) + rng.uniform(-1.0, 1.0, (3, 3))
        pos.append(rng.uniform(0, 1, (n, 3)) @ cell)
        cells.append(cell)
    ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)).to(DEV)
    P, L = torch.from_numpy(np.concatenate(pos)).to(DEV), torch.from_numpy(np.stack(cells)).to(DEV)
    for _ in range(2):
        out = ops.neighbor_list(P, L, ptr, r_max)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = ops.neighbor_list(P, L, ptr, r_max)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    from oracle.dos_oracle import neighbor_list_bruteforce
    t0 = time.perf_counter()
    for c in range(20):
        neighbor_list_bruteforce(pos[c], cells[c], r_max, True)
    t_cpu = (time.perf_counter() - t0) / 20 * C
    print(f"nlist {name:30s} C={C} atoms={int(sizes.sum())} edges={out['src'].numel()} r_max={r_max}: {t * 1e3:7.2f} ms for the whole "
          f"set (count + scan + fill, incl. 2 host reads) | numpy brute force ~{t_cpu * 1e3:8.0f} ms (extrapolated from 20)")


def node_mlp_case(name, M, H):
    """NodeModel MLP: the one-launch kernel (csrc/mlp2.hip) against the two dosx_gemm launches it replaces, fwd and bwd."""
    x, agg = torch.randn(M, H, device=DEV), torch.randn(M, H, device=DEV)
    w1, b1 = torch.randn(2 * H, 2 * H, device=DEV) / 16, torch.randn(2 * H, device=DEV)
    g, b, al = torch.randn(2 * H, device=DEV), torch.randn(2 * H, device=DEV), torch.full((1,), 0.25, device=DEV)
    w2, b2 = torch.randn(H, 2 * H, device=DEV) / 16, torch.randn(H, device=DEV)
    xhat, rstd, out = torch.empty(M, 2 * H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, H, device=DEV)
    us = timeit(lambda: ops.mlp_ln_fwd(M, x, agg, w1, b1, g, b, al, w2, b2, x, xhat, rstd, out, cs=False))
    cs_ok = ops.mlp_ln_cs(M, 2 * H, 2 * H, H)
    us_cs = us_cs3 = usb_cs = float("nan")
    if cs_ok:       # column-split form (round 6), alone and with the next layer's node products as its third phase
        w3n, pq = torch.randn(2 * H, 3 * H, device=DEV) / 16, torch.empty(M, 4 * H, device=DEV)
        us_cs = timeit(lambda: ops.mlp_ln_fwd(M, x, agg, w1, b1, g, b, al, w2, b2, x, xhat, rstd, out, cs=True))
        us_cs3 = timeit(lambda: ops.mlp_ln_fwd(M, x, agg, w1, b1, g, b, al, w2, b2, x, xhat, rstd, out, w3=w3n, nb3=2, pq=pq, cs=True))
    def two():
        ops.gemm(M, 2 * H, [ops.seg(x), ops.seg(agg)], w1, xhat, bias=b1, epi=ops.EPI_LN, aux_out=rstd)
        ops.gemm(M, H, [ops.seg(xhat)], w2, out, pro=ops.PRO_LN_PRELU, pro_gamma=g, pro_beta=b, pro_alpha=al, bias=b2, res=x)
    us2 = timeit(two)
    dy, dz, dcat = torch.randn(M, H, device=DEV), torch.empty(M, 2 * H, device=DEV), torch.empty(M, 2 * H, device=DEV)
    pld = 4 * H + 4
    part = torch.empty(ops.mlp_ln_bwd_partial_rows(M), pld, device=DEV)
    usb = timeit(lambda: ops.mlp_ln_bwd(M, dy, xhat, rstd, w1, w2, g, b, al, dz, dcat, part, cs=False))
    if cs_ok:
        usb_cs = timeit(lambda: ops.mlp_ln_bwd(M, dy, xhat, rstd, w1, w2, g, b, al, dz, dcat, part, cs=True))
    part2 = torch.empty(ops.gemm_partial_rows(M, 2 * H, ops.EPI_PRELU_LN_BWD), pld, device=DEV)
    def twob():
        ops.gemm(M, 2 * H, [ops.seg(dy)], w2, dz, w_layout=1, epi=ops.EPI_PRELU_LN_BWD, aux=xhat, aux_stats=rstd, epi_gamma=g,
                 epi_beta=b, epi_alpha=al, partials=part2, partial_ld=pld)
        ops.gemm(M, 2 * H, [ops.seg(dz)], w1, dcat, w_layout=1)
    usb2 = timeit(twob)
    fl = 2.0 * M * 2 * H * 3 * H
    print(f"nmlp  {name:30s} M={M} H={H}: fwd fused {us:6.1f} us ({fl / us / 1e6:5.1f} TF/s) | two GEMMs {us2:6.1f} us || "
          f"bwd fused {usb:6.1f} us ({fl / usb / 1e6:5.1f} TF/s) | two GEMMs {usb2:6.1f} us || column-split fwd {us_cs:6.1f} (+pq {us_cs3:6.1f}) bwd {usb_cs:6.1f} us"
          f"  (eager launches: the host's ~10 us per launch is the floor of these readings - kernel times: rocprofv3 --kernel-trace)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    args = ap.parse_args()
    w = args.what
    H = 128
    E, N, R1, R2 = 9000, 450, 51 * 64, 51 * 128
    if w in ("all", "ffn"):
        ffn_case("FFN fwd 2B", R2, H)
        ffn_case("FFN fwd B", R1, H)
        ffn_case("FFN fwd roofline scale", 262144, H)
        ffn_bwd_case("FFN bwd 2B", R2, H)
        ffn_bwd_case("FFN bwd B", R1, H)
        ffn_bwd_case("FFN bwd roofline scale", 262144, H)
    if w in ("all", "ffn", "layer"):
        layer_case("cross layer B (16-row tiles)", 51, 64, 12, 64, H)
        layer_case("cross layer 2B", 51, 128, 12, 64, H)
        layer_case("cross layer roofline scale", 51, 4096, 12, 2048, H)
        layer_case("self layer 2B (51 keys)", 51, 128, 51, 128, H)
        layer_case("self layer B (51 keys)", 51, 64, 51, 64, H)
    if w in ("all", "nmlp"):
        node_mlp_case("cfg2 nodes", N, H)
        node_mlp_case("one crystal", 7, H)
        node_mlp_case("cfg3 nodes (eDOS)", 1554, 256)
        node_mlp_case("4096 rows", 4096, H)
        node_mlp_case("roofline scale", 262144, H)
    if w in ("all", "neighbors"):
        neighbor_case("phonon-set sized", 1500, 4.0)
        neighbor_case("phonon-set sized", 1500, 6.0)
    if w in ("all", "collate"):
        collate_case("phonon 64 crystals", 64)
        collate_case("eDOS 64 crystals", 64, "edos")
    if w in ("all", "csr"):
        csr_case("cfg2 batch", 64)
        csr_case("cfg3 batch (eDOS)", 64, "edos")
        csr_case("512 crystals", 512)
    if w in ("all", "gemm"):
        gemm_case("edge gemm1 (gather, LN epi)", E, 2 * H, 3 * H, epi=ops.EPI_LN, gather=True)
        gemm_case("edge gemm1 (plain A, LN epi)", E, 2 * H, 3 * H, epi=ops.EPI_LN)
        gemm_case("edge gemm2 (LN_PRELU pro)", E, H, 2 * H, pro=ops.PRO_LN_PRELU)
        gemm_case("edge dgrad (dz->dcat)", E, 3 * H, 2 * H, wl=1)
        gemm_case("edge dgrad2 (dy->da)", E, 2 * H, H, wl=1)
        gemm_case("node gemm1", N, 2 * H, 2 * H, epi=ops.EPI_LN)
        gemm_case("ffn fc1 (rowLN pro) 2B", R2, 4 * H, H, pro=ops.PRO_ROWLN)
        gemm_case("ffn fc2 2B", R2, H, 4 * H)
        gemm_case("ffn fc1 B", R1, 4 * H, H, pro=ops.PRO_ROWLN)
        gemm_case("ffn dgrad fc2 (dy->dh)", R2, 4 * H, H, wl=1)
        gemm_case("ffn dgrad fc1 (dh->dx)", R2, H, 4 * H, wl=1)
        gemm_case("big  (roofline scale)", 262144, 512, 128, pro=0)
        gemm_case("big2 (roofline scale)", 262144, 256, 384, epi=ops.EPI_LN)
        gemm_case("eDOS edge gemm1 H256", 16000, 512, 768, epi=ops.EPI_LN)
        gemm_case("eDOS fc1 H256 2B", 201 * 128, 1024, 256, pro=ops.PRO_ROWLN)
        gemm_case("eDOS edge da (PRELU_LN_BWD epi)", 17880, 512, 256, wl=1, epi=ops.EPI_PRELU_LN_BWD)
        gemm_case("cfg2 edge da (PRELU_LN_BWD epi)", 9000, 256, 128, wl=1, epi=ops.EPI_PRELU_LN_BWD)
        gemm_case("node encoder dz (PRELU_BWD epi)", 424, 128, 128, wl=1, epi=ops.EPI_PRELU_BWD)
        gemm_case("edge encoder dz (PRELU_BWD epi)", 9344, 128, 128, wl=1, epi=ops.EPI_PRELU_BWD)
    if w in ("all", "bf16x3"):     # split-bf16 next to the exact-fp32 kernel: the Electron-DOS feed-forward shapes + roofline scale
        for name, M_, N_, K_, wl_ in (("eDOS fc1 fwd", 25728, 1024, 256, 0), ("eDOS fc2 dgrad", 25728, 1024, 256, 1),
                                      ("eDOS fc2 fwd", 25728, 256, 1024, 0), ("eDOS fc1 dgrad", 25728, 256, 1024, 1),
                                      ("roofline scale", 262144, 512, 128, 0), ("square 8192", 8192, 8192, 8192, 0)):
            bf16x3_case(name, M_, N_, K_, wl_)
    if w == "edosffn":      # the four feed-forward GEMMs of the Electron-DOS step (tile-policy experiments: DOSX_GEMM_RT / _BN)
        for M_ in (201 * 128, 24576, 201 * 64):
            gemm_case("eDOS fc1 fwd (rowLN pro)", M_, 1024, 256, pro=ops.PRO_ROWLN)
            gemm_case("eDOS fc1 fwd, plain A", M_, 1024, 256)
            gemm_case("eDOS fc2 fwd", M_, 256, 1024)
            gemm_case("eDOS fc2 dgrad (dy->dh)", M_, 1024, 256, wl=1)
            gemm_case("eDOS fc1 dgrad (dh->dx)", M_, 256, 1024, wl=1)
    if w in ("all", "wgrad"):
        wgrad_case("edge W1 (2H x 3H)", E, 2 * H, 3 * H)
        wgrad_case("edge W2 (H x 2H)", E, H, 2 * H)
        wgrad_case("fc1 (4H x H) 2B", R2, 4 * H, H)
        wgrad_case("fc2 (H x 4H) 2B", R2, H, 4 * H)
        wgrad_case("node W1", N, 2 * H, 2 * H)
        wgrad_case("big", 262144, 512, 128)
        # the four weight shapes of a model with hidden 128 at roofline scale
        wgrad_case("big edge W1 (2H x 3H)", 262144, 256, 384)
        wgrad_case("big edge W2 (H x 2H)", 262144, 128, 256)
        wgrad_case("big fc1 (4H x H)", 262144, 512, 128)
        wgrad_case("big fc2 (H x 4H)", 262144, 128, 512)
    if w in ("all", "scatter"):
        segreduce_case("cfg2 batch", 450, 20, 128)
        segreduce_case("cfg2 batch (last layer)", 450, 20, 128, residual=False)
        segreduce_case("roofline scale 4Mi edges", 209715, 20, 128)
        segreduce_case("roofline scale, no residual", 209715, 20, 128, residual=False)
        segreduce_case("roofline scale H=256", 104857, 20, 256)
    if w in ("all", "attn"):
        attn_case("phonon cross (energies->atoms)", 51, 64, 12, 64, 128)
        attn_case("phonon cross 2B", 51, 128, 12, 64, 128)
        attn_case("phonon self 2B", 51, 128, 51, 128, 128)
        attn_case("eDOS cross B", 201, 64, 41, 64, 256)
        attn_case("eDOS cross 2B", 201, 128, 41, 64, 256)
        attn_case("eDOS cross 2B, 64 atoms", 201, 128, 64, 64, 256)
        attn_case("eDOS self 2B", 201, 128, 201, 128, 256)
        attn_case("roofline scale self", 201, 2048, 201, 2048, 256)


if __name__ == "__main__":
    main()
