"""Kernel-level parity: each libdosx entry point (through the C ABI / ctypes) against a plain
torch float64 reference of the same op, on seeded inputs.  Needs a real MI355X."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from dostransformer_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(torch.float32).to(DEV)


def err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


TOL = 2e-5


def prelu(x, a):
    return torch.where(x >= 0, x, a * x)


@pytest.mark.parametrize("M,N,K", [(100, 128, 64), (33, 256, 384), (257, 512, 128), (64, 64, 118), (5, 16, 41),
                                    (1000, 512, 100), (77, 32, 4),
                                    # 48-row tiles (8192 < M <= 12288, one column tile): aligned, ragged, unaligned K
                                    (9000, 128, 256), (10001, 96, 41), (12288, 128, 118), (8193, 256, 64),
                                    # 32- and 64-row tiles with ragged edges
                                    (5000, 100, 64), (20011, 128, 96),
                                    # two launches: the full rounds of 64-row tiles, then the tail rows as 16- / 48-row / 64-row tiles
                                    # (gemm_tail_split; 20011 above: 16384 + 3627 rows)
                                    (25728, 256, 128), (25728, 1024, 64), (16384 + 9001, 128, 41), (12864, 1024, 32)])
@pytest.mark.parametrize("wl", [0, 1])
def test_gemm_plain(M, N, K, wl):
    o = ops()
    a = rnd(M, K, seed=1)
    w = rnd(N, K, seed=2) if wl == 0 else rnd(K, N, seed=2)
    b = rnd(N, seed=3)
    out = torch.empty(M, N, device=DEV)
    o.gemm(M, N, [o.seg(a)], w, out, w_layout=wl, bias=b)
    ref = a.double() @ (w.double().T if wl == 0 else w.double()) + b.double()
    assert err(out, ref) < TOL


def test_gemm_asymmetric_identity():
    # A = I with an asymmetric B catches a transposed C write (guide: "A=I-check with ASYMMETRIC B")
    o = ops()
    n = 64
    a = torch.eye(n, device=DEV)
    w = (torch.arange(n * n, device=DEV, dtype=torch.float32).reshape(n, n) % 97) / 7.0
    out = torch.empty(n, n, device=DEV)
    o.gemm(n, n, [o.seg(a)], w, out, w_layout=0)
    assert torch.equal(out, w.T.contiguous())
    o.gemm(n, n, [o.seg(a)], w, out, w_layout=1)
    assert torch.equal(out, w)


def test_gemm_gather_concat_act_residual_remap():
    o = ops()
    n, e, h = 50, 333, 64
    x = rnd(n, h, seed=1)
    ea = rnd(e, h, seed=2)
    src = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(DEV)
    dst = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(4)).to(torch.int32).to(DEV)
    w = rnd(128, 3 * h, seed=5, scale=0.1)
    b = rnd(128, seed=6)
    res = rnd(e, 128, seed=7)
    out = torch.empty(e, 128, device=DEV)
    segs = [o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(ea)]
    o.gemm(e, 128, segs, w, out, bias=b, act=o.ACT_LEAKY, act_slope=0.01, res=res)
    cat = torch.cat([x[src.long()], x[dst.long()], ea], 1).double()
    ref = F.leaky_relu(cat @ w.double().T + b.double(), 0.01) + res.double()
    assert err(out, ref) < TOL
    # DIV / MOD maps and remapped output rows: rows r = s*B + b
    S, B = 7, 5
    en = rnd(S, h, seed=8)
    gr = rnd(B, h, seed=9)
    w2 = rnd(h, 2 * h, seed=10, scale=0.1)
    out2 = torch.zeros(S * 2 * B, h, device=DEV)
    segs = [o.seg(en, rmap=o.rowmap(d=B, m=1, c=0)), o.seg(gr, rmap=o.rowmap(d=B, m=0, c=1))]
    o.gemm(S * B, h, segs, w2, out2, out_map=o.rowmap(d=B, m=2 * B, c=1, off=B))
    cat = torch.cat([en[:, None, :].expand(S, B, h), gr[None].expand(S, B, h)], 2).reshape(S * B, 2 * h).double()
    ref = (cat @ w2.double().T).reshape(S, B, h)
    got = out2.reshape(S, 2 * B, h)
    assert err(got[:, B:], ref) < TOL and float(got[:, :B].abs().max()) == 0.0


# small M runs the 16-row (HALF) tiles, 4500 the 32-row and 9000 the 64-row ones (gemm_rt in csrc/gemm.hip)
@pytest.mark.parametrize("M,H2,K", [(100, 256, 384), (37, 512, 768), (64, 128, 192), (9, 32, 48), (300, 16, 24),
                                    (4500, 256, 384), (9000, 256, 384)])
def test_gemm_ln_epilogue_and_ln_prelu_prologue(M, H2, K):
    o = ops()
    a = rnd(M, K, seed=1)
    w = rnd(H2, K, seed=2, scale=0.2)
    b = rnd(H2, seed=3)
    xhat = torch.empty(M, H2, device=DEV)
    rstd = torch.empty(M, device=DEV)
    o.gemm(M, H2, [o.seg(a)], w, xhat, bias=b, epi=o.EPI_LN, aux_out=rstd)
    z = a.double() @ w.double().T + b.double()
    mu, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    ref = (z - mu) / torch.sqrt(var + 1e-5)
    assert err(xhat, ref) < 5e-5
    assert err(rstd, (1 / torch.sqrt(var + 1e-5)).squeeze(1)) < 5e-5
    # second linear with LN-affine + PReLU prologue and residual
    gam, bet = rnd(H2, seed=4), rnd(H2, seed=5)
    alpha = torch.tensor([0.25], device=DEV)
    w3 = rnd(H2 // 2, H2, seed=6, scale=0.2)
    b3 = rnd(H2 // 2, seed=7)
    res = rnd(M, H2 // 2, seed=8)
    y = torch.empty(M, H2 // 2, device=DEV)
    o.gemm(M, H2 // 2, [o.seg(xhat)], w3, y, pro=o.PRO_LN_PRELU, pro_gamma=gam, pro_beta=bet, pro_alpha=alpha,
           bias=b3, res=res)
    act = prelu(xhat.double() * gam.double() + bet.double(), 0.25)
    assert err(y, act @ w3.double().T + b3.double() + res.double()) < TOL


@pytest.mark.parametrize("M", [200, 3264, 6528])
def test_gemm_rowln_prologue_and_stats_out(M):
    o = ops()
    H = 128
    x = rnd(M, H, seed=1)
    w1 = rnd(4 * H, H, seed=2, scale=0.1)
    b1 = rnd(4 * H, seed=3)
    gam, bet = rnd(H, seed=4), rnd(H, seed=5)
    mu = x.double().mean(1)
    rs = 1 / torch.sqrt(x.double().var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mu, rs], 1).float().contiguous()
    h = torch.empty(M, 4 * H, device=DEV)
    o.gemm(M, 4 * H, [o.seg(x)], w1, h, pro=o.PRO_ROWLN, pro_gamma=gam, pro_beta=bet, pro_stats=stats, bias=b1,
           act=o.ACT_RELU)
    ln = F.layer_norm(x.double(), (H,), gam.double(), bet.double(), 1e-5)
    ref = F.relu(ln @ w1.double().T + b1.double())
    assert err(h, ref) < TOL
    w2 = rnd(H, 4 * H, seed=6, scale=0.1)
    b2 = rnd(H, seed=7)
    x2 = torch.empty(M, H, device=DEV)
    st2 = torch.empty(M, 2, device=DEV)
    o.gemm(M, H, [o.seg(h)], w2, x2, bias=b2, res=x, stats_out=st2)
    ref2 = x.double() + ref @ w2.double().T + b2.double()
    assert err(x2, ref2) < TOL
    assert err(st2[:, 0], ref2.mean(1)) < 5e-5
    assert err(st2[:, 1], 1 / torch.sqrt(ref2.var(1, unbiased=False) + 1e-5)) < 5e-5


def _reduce(o, sink):
    sink.flush()
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(1000, 256, 384), (333, 128, 118), (70, 64, 64), (5000, 128, 512), (40, 16, 24)])
def test_wgrad_and_reduce(M, N, K):
    o = ops()
    dy = rnd(M, N, seed=1)
    a = rnd(M, K, seed=2)
    ns = o.wgrad_splits(M, N, K)
    sink = o.GradSink(DEV)
    slab = sink.scratch(ns, N, K)
    slab_b = sink.scratch(ns, N)
    o.wgrad(M, N, o.seg(dy), [o.seg(a)], slab, slab_b, ns)
    dw = torch.empty(N, K, device=DEV)
    db = torch.empty(N, device=DEV)
    sink.add(slab, 0, dw, ns, N * K, N * K)
    sink.add(slab_b, 0, db, ns, N, N)
    _reduce(o, sink)
    assert err(dw, dy.double().T @ a.double()) < TOL
    assert err(db, dy.double().sum(0)) < TOL
    # duplicate destination -> accumulates in a second wave
    sink.add(slab, 0, dw, ns, N * K, N * K)
    sink.add(slab, 0, dw, ns, N * K, N * K)
    _reduce(o, sink)
    assert err(dw, 2 * (dy.double().T @ a.double())) < TOL


def test_wgrad_prologues_and_gather():
    o = ops()
    n, e, h = 40, 500, 32
    x = rnd(n, h, seed=1)
    ea = rnd(e, h, seed=2)
    src = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(DEV)
    dst = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(4)).to(torch.int32).to(DEV)
    dz = rnd(e, 2 * h, seed=5)
    segs = [o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(ea)]
    ns = o.wgrad_splits(e, 2 * h, 3 * h)
    sink = o.GradSink(DEV)
    slab = sink.scratch(ns, 2 * h, 3 * h)
    o.wgrad(e, 2 * h, o.seg(dz), segs, slab, None, ns)
    cat = torch.cat([x[src.long()], x[dst.long()], ea], 1).double()
    assert err(slab.sum(0), dz.double().T @ cat) < TOL
    # LN_PRELU prologue
    xhat = rnd(e, 2 * h, seed=6)
    gam, bet = rnd(2 * h, seed=7), rnd(2 * h, seed=8)
    alpha = torch.tensor([0.3], device=DEV)
    dy = rnd(e, h, seed=9)
    ns = o.wgrad_splits(e, h, 2 * h)
    slab = sink.scratch(ns, h, 2 * h)
    o.wgrad(e, h, o.seg(dy), [o.seg(xhat)], slab, None, ns, pro=o.PRO_LN_PRELU, pro_gamma=gam, pro_beta=bet,
            pro_alpha=alpha)
    act = prelu(xhat.double() * gam.double() + bet.double(), 0.3)
    assert err(slab.sum(0), dy.double().T @ act) < TOL


@pytest.mark.parametrize("M,N,K", [(15, 64, 16), (40, 64, 16), (100, 128, 32), (6528, 512, 128), (33, 16, 64)])
def test_wgrad_rowln_prologue(M, N, K):
    """dW of fc1 with the pre-FFN LayerNorm recomputed from saved (mean, rstd) (transformer.py:141-143)."""
    o = ops()
    dy, x, gam, bet = rnd(M, N, seed=1), rnd(M, K, seed=2), rnd(K, seed=3), rnd(K, seed=4)
    mu = x.mean(1, keepdim=True)
    rstd = 1 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    stats = torch.cat([mu, rstd], 1).contiguous()
    ns = o.wgrad_splits(M, N, K)
    slab = torch.empty(ns, N, K, device=DEV)
    slab_b = torch.empty(ns, N, device=DEV)
    o.wgrad(M, N, o.seg(dy), [o.seg(x)], slab, slab_b, ns, pro=o.PRO_ROWLN, pro_gamma=gam, pro_beta=bet, pro_stats=stats)
    ref = dy.double().T @ (((x - mu) * rstd) * gam + bet).double()
    assert err(slab.sum(0), ref) < TOL
    assert err(slab_b.sum(0), dy.double().sum(0)) < TOL


def test_wgrad_gather_fast_path():
    """cat[x[row], x[col], e] with 64-wide segments: every K tile lies inside one segment (buffer-addressed staging)."""
    o = ops()
    n, e, h = 40, 1000, 64
    x, ea = rnd(n, h, seed=1), rnd(e, h, seed=2)
    src = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(DEV)
    dst = torch.randint(0, n, (e,), generator=torch.Generator().manual_seed(4)).to(torch.int32).to(DEV)
    dz = rnd(e, 2 * h, seed=5)
    segs = [o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(ea)]
    ns = o.wgrad_splits(e, 2 * h, 3 * h)
    slab = torch.empty(ns, 2 * h, 3 * h, device=DEV)
    o.wgrad(e, 2 * h, o.seg(dz), segs, slab, None, ns)
    cat = torch.cat([x[src.long()], x[dst.long()], ea], 1).double()
    assert err(slab.sum(0), dz.double().T @ cat) < TOL


@pytest.mark.parametrize("M,H2", [(100, 256), (45, 512), (33, 32), (4500, 256), (9000, 256), (25728, 256), (17880, 512), (24576 + 700, 128),
                                  (8940, 512), (8192 + 33, 512)])       # (512 columns: 32-row tiles + a tail of 16-row tiles, round 5)
def test_gemm_prelu_ln_bwd_epilogue(M, H2):
    o = ops()
    H = H2 // 2
    z = rnd(M, H2, seed=1).double().requires_grad_(True)
    gam = rnd(H2, seed=2).double().requires_grad_(True)
    bet = rnd(H2, seed=3).double().requires_grad_(True)
    alpha = torch.tensor([0.25], device=DEV, dtype=torch.float64, requires_grad=True)
    w3 = rnd(H, H2, seed=4, scale=0.2)
    dy = rnd(M, H, seed=5)
    ln = F.layer_norm(z, (H2,), gam, bet, 1e-5)
    y = prelu(ln, alpha) @ w3.double().T
    y.backward(dy.double())
    mu, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    xhat = ((z - mu) / torch.sqrt(var + 1e-5)).detach().float().contiguous()
    rstd = (1 / torch.sqrt(var + 1e-5)).detach().float().reshape(-1).contiguous()
    dz = torch.empty(M, H2, device=DEV)
    rows = o.gemm_partial_rows(M, H2, o.EPI_PRELU_LN_BWD)
    part = torch.full((rows, 2 * H2 + 1), float('nan'), device=DEV)
    o.gemm(M, H2, [o.seg(dy)], w3, dz, w_layout=1, epi=o.EPI_PRELU_LN_BWD, aux=xhat, aux_stats=rstd,
           epi_gamma=gam.detach().float(), epi_beta=bet.detach().float(), epi_alpha=alpha.detach().float(),
           partials=part, partial_ld=2 * H2 + 1)
    assert err(dz, z.grad) < 5e-5
    ps = part.double().sum(0)
    assert err(ps[:H2], gam.grad) < 5e-5
    assert err(ps[H2:2 * H2], bet.grad) < 5e-5
    assert abs(float(ps[2 * H2]) - float(alpha.grad)) < 5e-5 * (1 + abs(float(alpha.grad)))


@pytest.mark.parametrize("M", [150, 5000, 13000, 25728])
def test_gemm_rowln_bwd_relu_mask_prelu_bwd(M):
    o = ops()
    H = 64
    x = rnd(M, H, seed=1).double().requires_grad_(True)
    gam = rnd(H, seed=2).double().requires_grad_(True)
    bet = rnd(H, seed=3).double().requires_grad_(True)
    w1 = rnd(4 * H, H, seed=4, scale=0.2)
    dh = rnd(M, 4 * H, seed=5)
    res = rnd(M, H, seed=6)
    (F.layer_norm(x, (H,), gam, bet, 1e-5) @ w1.double().T).backward(dh.double())
    mu = x.detach().mean(1)
    rs = 1 / torch.sqrt(x.detach().var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mu, rs], 1).float().contiguous()
    dx = torch.empty(M, H, device=DEV)
    rows = o.gemm_partial_rows(M, H, o.EPI_ROWLN_BWD)
    part = torch.full((rows, 2 * H), float('nan'), device=DEV)
    o.gemm(M, H, [o.seg(dh)], w1, dx, w_layout=1, epi=o.EPI_ROWLN_BWD, aux=x.detach().float(), aux_stats=stats,
           epi_gamma=gam.detach().float(), res=res, partials=part, partial_ld=2 * H)
    assert err(dx, x.grad + res.double()) < 5e-5
    ps = part.double().sum(0)
    assert err(ps[:H], gam.grad) < 5e-5 and err(ps[H:], bet.grad) < 5e-5
    # relu mask
    hsaved = rnd(M, 4 * H, seed=7)
    w2 = rnd(H, 4 * H, seed=8, scale=0.2)
    dy = rnd(M, H, seed=9)
    out = torch.empty(M, 4 * H, device=DEV)
    o.gemm(M, 4 * H, [o.seg(dy)], w2, out, w_layout=1, epi=o.EPI_RELU_MASK, aux=hsaved)
    assert err(out, (dy.double() @ w2.double()) * (hsaved > 0)) < TOL
    # prelu bwd (N tiled by 128: two partial columns per row block)
    z = rnd(M, 4 * H, seed=10)
    alpha = torch.tensor([0.2], device=DEV)
    rows = o.gemm_partial_rows(M, 4 * H, o.EPI_PRELU_BWD)
    part = torch.full((rows, 1), float('nan'), device=DEV)
    o.gemm(M, 4 * H, [o.seg(dy)], w2, out, w_layout=1, epi=o.EPI_PRELU_BWD, aux=z, epi_alpha=alpha, partials=part,
           partial_ld=1)
    da = dy.double() @ w2.double()
    assert err(out, torch.where(z >= 0, da, 0.2 * da)) < TOL
    assert abs(float(part.double().sum()) - float((da * z.double() * (z < 0)).sum())) < 1e-3


def test_edge_features():
    o = ops()
    from oracle import dos_oracle as O
    v = rnd(1000, 3, seed=1, scale=2.0)
    v[0] = 0
    v[1] = torch.tensor([4.0, 0, 0])
    got = o.edge_feat_sh1(v, 4.0)
    assert err(got, O.edge_features_sh1(v.double().cpu()).to(DEV)) < 1e-5


@pytest.mark.parametrize("H", [128, 256, 16, 64])
@pytest.mark.parametrize("mean", [True, False])
def test_segment_reduce_and_backward(H, mean):
    o = ops()
    from dostransformer_amd import synth
    from dostransformer_amd.batch import graph_meta
    g = synth.phonon_batch(5, seed=3, dtype=torch.float32)
    m = graph_meta(g, DEV)
    N, E = m.num_nodes, m.num_edges
    msg = rnd(E, H, seed=1)
    e_in = rnd(E, H, seed=2)
    agg = torch.empty(N, H, device=DEV)
    e_out = torch.empty(E, H, device=DEV)
    o.segment_reduce(msg, m.rowptr_dst, m.inv_deg if mean else None, agg, e_in, e_out, N, E, H)
    ref = torch.zeros(N, H, device=DEV, dtype=torch.float64).index_add_(0, m.dst.long(), msg.double())
    if mean:
        ref = ref * m.inv_deg.double()[:, None]
    assert err(agg, ref) < 1e-5
    assert err(e_out, e_in.double() + msg.double()) < 1e-6
    # edge grad combine
    dcat_n = rnd(N, 2 * H, seed=3)
    de_new = rnd(E, H, seed=4)
    dmsg = torch.empty(E, H, device=DEV)
    o.edge_grad_combine(de_new, dcat_n.data_ptr() + 4 * H, 2 * H, m.dst, m.inv_deg if mean else None, dmsg, E, H)
    sc = m.inv_deg.double()[m.dst.long()][:, None] if mean else 1.0
    assert err(dmsg, de_new.double() + dcat_n[:, H:].double()[m.dst.long()] * sc) < 1e-6
    # gather backward
    dcat = rnd(E, 3 * H, seed=5)
    dx_res = rnd(N, H, seed=6)
    dx = torch.empty(N, H, device=DEV)
    de_out = torch.empty(E, H, device=DEV)
    o.gather_bwd(dcat, dcat_n.data_ptr(), 2 * H, dx_res, m.rowptr_dst, m.rowptr_src, m.perm_src, de_new, dx, de_out,
                 N, E, H)
    ref = dx_res.double() + dcat_n[:, :H].double()
    ref = ref.index_add(0, m.dst.long(), dcat[:, H:2 * H].double()).index_add(0, m.src.long(), dcat[:, :H].double())
    assert err(dx, ref) < 1e-5
    assert err(de_out, de_new.double() + dcat[:, 2 * H:].double()) < 1e-6


def test_pool_dense_norm():
    o = ops()
    from dostransformer_amd import synth
    from dostransformer_amd.batch import graph_meta
    g = synth.phonon_batch(6, seed=4, dtype=torch.float32)
    m = graph_meta(g, DEV)
    N, B, H, nmax = m.num_nodes, m.num_graphs, 128, m.n_max
    x = rnd(N, H, seed=1)
    pooled = torch.empty(B, H, device=DEV)
    o.graph_pool(x, m.graph_ptr, pooled.data_ptr(), H, B, H)
    ref = torch.zeros(B, H, device=DEV, dtype=torch.float64).index_add_(0, m.node_graph.long(), x.double())
    assert err(pooled, ref) < 1e-5
    dx = rnd(N, H, seed=2)
    dx0 = dx.clone()
    dp = rnd(B, H, seed=3)
    o.graph_pool_bwd(dp.data_ptr(), H, m.node_graph, dx, N, H, True)
    assert err(dx, dx0.double() + dp.double()[m.node_graph.long()]) < 1e-6
    kv = torch.full((nmax * B, H), 7.0, device=DEV)
    rstd = torch.empty(N, device=DEV)
    o.dense_normalize(x, m.dense_row, kv, rstd, N, H, nmax * B)
    xr = x.double().requires_grad_(True)
    xh = F.layer_norm(xr, (H,), None, None, 1e-5)
    ref = torch.zeros(nmax * B, H, device=DEV, dtype=torch.float64)
    ref[m.dense_row.long()] = xh.detach()
    assert err(kv, ref) < 1e-5
    dkv = rnd(nmax * B, H, seed=5)
    xh.backward(dkv.double()[m.dense_row.long()])
    dxn = torch.zeros(N, H, device=DEV)
    o.dense_normalize_bwd(dkv, kv, rstd, m.dense_row, dxn, N, H, False)
    assert err(dxn, xr.grad) < 5e-5


def test_layernorm_rowdot():
    o = ops()
    S, Bq, H = 51, 6, 128
    M = S * Bq
    x = rnd(M, H, seed=1).double().requires_grad_(True)
    gam = rnd(H, seed=2).double().requires_grad_(True)
    bet = rnd(H, seed=3).double().requires_grad_(True)
    w = rnd(H, seed=4).double().requires_grad_(True)
    b = rnd(1, seed=5).double().requires_grad_(True)
    y = F.layer_norm(x, (H,), gam, bet, 1e-5)
    dos_ref = (y @ w + b).reshape(S, Bq).T
    ddos = rnd(Bq, S, seed=6)
    dos_ref.backward(ddos.double())
    f = lambda t: t.detach().float().contiguous()
    xhat = torch.empty(M, H, device=DEV)
    rstd = torch.empty(M, device=DEV)
    dos = torch.empty(Bq, S, device=DEV)
    o.ln_rowdot(f(x), f(gam), f(bet), f(w), f(b), xhat, rstd, dos, S, Bq, H)
    assert err(dos, dos_ref) < 2e-5
    dx = torch.empty(M, H, device=DEV)
    rows = (M + 31) // 32
    part = torch.empty(rows, 3 * H + 1, device=DEV)
    o.ln_rowdot_bwd(ddos, xhat, rstd, f(gam), f(bet), f(w), dx, part, S, Bq, H)
    ps = part.double().sum(0)
    assert err(dx, x.grad) < 5e-5
    assert err(ps[:H], gam.grad) < 5e-5 and err(ps[H:2 * H], bet.grad) < 5e-5
    assert err(ps[2 * H:3 * H], w.grad) < 5e-5 and abs(float(ps[3 * H]) - float(b.grad)) < 1e-3
    # plain LN fwd/bwd
    yout = torch.empty(M, H, device=DEV)
    o.layernorm(f(x), f(gam), f(bet), yout, xhat, rstd, M, H)
    assert err(yout, y) < 2e-5
    x.grad = None
    gam.grad = None
    bet.grad = None
    dy = rnd(M, H, seed=7)
    F.layer_norm(x, (H,), gam, bet, 1e-5).backward(dy.double())
    part = torch.empty(rows, 2 * H, device=DEV)
    o.layernorm_bwd(dy, xhat, rstd, f(gam), dx, part, M, H)
    ps = part.double().sum(0)
    assert err(dx, x.grad) < 5e-5 and err(ps[:H], gam.grad) < 5e-5 and err(ps[H:], bet.grad) < 5e-5


def _attn_ref(x, kvhat, gam, bet, Sq, Bq, Nk, Bk, H, qs, qb, mask=None):
    """float64 torch reference of the attention block (same math as oracle.encoder_layer's first half); mask: the
    attention-dropout multiplier applied to the softmax output (multihead_attention.py:70)."""
    rows = (torch.arange(Sq, device=DEV)[:, None] * qs + torch.arange(Bq, device=DEV)[None, :] * qb).reshape(-1)
    xq = x[rows].reshape(Sq, Bq, H)
    q = F.layer_norm(xq, (H,), gam, bet, 1e-5)
    k = (kvhat * gam + bet).reshape(Nk, Bk, H)
    k = k[:, torch.arange(Bq, device=DEV) % Bk]
    w = torch.bmm(q.transpose(0, 1), k.permute(1, 2, 0)) * H ** -0.5
    p = torch.softmax(w, -1)
    pd = p if mask is None else p * mask
    out = xq + torch.bmm(pd, k.transpose(0, 1)).transpose(0, 1)
    return out.reshape(Sq * Bq, H), p


@pytest.mark.parametrize("Sq,Bq,Nk,Bk,H,bcast", [(51, 4, 12, 4, 128, False), (51, 6, 9, 3, 64, False),
                                                   (201, 2, 41, 2, 256, False), (70, 3, 70, 3, 128, False),
                                                   (51, 5, 7, 5, 128, True), (201, 2, 201, 2, 256, False),
                                                   (7, 3, 5, 3, 16, False),
                                                   # more than 320 keys: the general kernels behind the same contract
                                                   (51, 4, 330, 2, 128, False), (20, 3, 700, 3, 64, False),
                                                   (51, 4, 321, 4, 128, True), (9, 2, 400, 2, 256, False)])
@pytest.mark.parametrize("drop", [0.0, 0.35])
@pytest.mark.parametrize("pkv", [False, True])
def test_attention_fwd_bwd(Sq, Bq, Nk, Bk, H, bcast, pkv, drop):
    """pkv: the dq kernel also produces the per-tile dK + dV partials and a reduction kernel finishes the key gradient
    (Nk <= 64; DosxAttn.dkv_part) instead of the streamed dkv kernel behind the dscores round trip."""
    from dostransformer_amd import _lib
    if pkv and not _lib.load().dosx_attention_pkv_supported(Nk, H):
        pytest.skip("partial-dKV path covers Nk <= 64")
    o = ops()
    from dostransformer_amd._lib import Attn
    qs, qb = (1, 0) if bcast else (Bq, 1)
    xrows = Sq if bcast else Sq * Bq
    x = rnd(xrows, H, seed=1).double().requires_grad_(True)
    kv = rnd(Nk * Bk, H, seed=2)
    kv[-Bk:] = 0            # zero-padded atoms: LN gives beta on them
    kv = kv.double().requires_grad_(True)
    gam = rnd(H, seed=3).double().requires_grad_(True)
    bet = (0.3 * rnd(H, seed=4)).double().requires_grad_(True)
    mask = None
    if drop > 0:          # an arbitrary Bernoulli multiplier: the kernels take the mask as an operand
        mask = (torch.rand(Bq, Sq, Nk, generator=torch.Generator().manual_seed(9)) >= drop).float().to(DEV) / (1 - drop)
    ref, pref = _attn_ref(x, kv, gam, bet, Sq, Bq, Nk, Bk, H, qs, qb, None if mask is None else mask.double())
    dout = rnd(Sq * Bq, H, seed=5)
    ref.backward(dout.double())
    f = lambda t: t.detach().float().contiguous()
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, qs, qb
    xf, kvf, gf, bf = f(x), f(kv), f(gam), f(bet)
    out = torch.empty(Sq * Bq, H, device=DEV)
    probs = torch.empty(Bq, Sq, Nk, device=DEV)
    qstats = torch.empty(Sq * Bq, 2, device=DEV)
    ostats = torch.empty(Sq * Bq, 2, device=DEV)
    a.x, a.kvhat, a.gamma0, a.beta0 = xf.data_ptr(), kvf.data_ptr(), gf.data_ptr(), bf.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qstats.data_ptr(), ostats.data_ptr()
    a.drop_mask = mask.data_ptr() if mask is not None else None
    o.attention_fwd(a)
    assert err(out, ref) < 3e-5
    assert err(probs, pref) < 3e-5
    assert err(ostats[:, 0], ref.detach().mean(1)) < 5e-5
    assert err(ostats[:, 1], 1 / torch.sqrt(ref.detach().var(1, unbiased=False) + 1e-5)) < 5e-5
    dx = torch.empty(Sq * Bq, H, device=DEV)
    dsc = torch.empty(Bq, Sq, Nk, device=DEV)
    dkv = torch.zeros(Nk * Bk, H, device=DEV)
    nqt, nkt = (Sq + 31) // 32, ((Nk + 15) // 16 if pkv else (Nk + 31) // 32)
    part = torch.empty(Bq * nqt + Bk * nkt, 2 * H, device=DEV)
    a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), dsc.data_ptr(), dkv.data_ptr(), 1
    a.partials_q = part.data_ptr()
    a.partials_kv = part.data_ptr() + 4 * Bq * nqt * 2 * H
    if pkv:
        kvp = torch.full((Bq * nqt * Nk, H), float("nan"), device=DEV)
        a.dkv_part, a.dscores = kvp.data_ptr(), None
        dkv += 0.5                           # accumulate flag: the reduction adds onto what is there
    o.attention_bwd(a)
    if pkv:
        dkv -= 0.5
    ps = part.double().sum(0)
    if bcast:
        dxr = dx.double().reshape(Sq, Bq, H).sum(1)
    else:
        dxr = dx
    assert err(dxr, x.grad) < 5e-5
    assert err(dkv, kv.grad) < 5e-5
    assert err(ps[:H], gam.grad) < 5e-5 and err(ps[H:], bet.grad) < 5e-5


def test_losses_adamw_misc():
    o = ops()
    B, S = 7, 51
    pg, ps_, y = rnd(B, S, seed=1), rnd(B, S, seed=2), rnd(B, S, seed=3).abs()
    sse = torch.empty(2, device=DEV)
    o.sse2(pg, ps_, y, sse, B * S)
    a = pg.double().requires_grad_(True)
    b = ps_.double().requires_grad_(True)
    loss = torch.sqrt(F.mse_loss(a, y.double())) + 0.7 * torch.sqrt(F.mse_loss(b, y.double()))
    loss.backward()
    dpg, dps, l = torch.empty(B, S, device=DEV), torch.empty(B, S, device=DEV), torch.empty(1, device=DEV)
    o.loss_phonon_bwd(pg, ps_, y, sse, 0.7, B * S, dpg, dps, l, B * S)
    assert abs(float(l) - float(loss)) < 1e-5 and err(dpg, a.grad) < 1e-5 and err(dps, b.grad) < 1e-5
    S = 201
    pg, ps_, yft = rnd(B, S, seed=4), rnd(B, S, seed=5), rnd(B * S, seed=6)
    a = pg.double().requires_grad_(True)
    b = ps_.double().requires_grad_(True)
    yy = torch.where(yft < 0, torch.zeros_like(yft), yft).double().reshape(B, S)
    loss = torch.sqrt(((yy - a) ** 2).mean(1)).mean() + 0.5 * torch.sqrt(((yy - b) ** 2).mean(1)).mean()
    loss.backward()
    dpg, dps, lp = torch.empty(B, S, device=DEV), torch.empty(B, S, device=DEV), torch.empty(B, device=DEV)
    o.loss_edos(pg, ps_, yft, 0.5, B, S, B, dpg, dps, lp)
    assert abs(float(lp.sum()) - float(loss)) < 1e-5 and err(dpg, a.grad) < 1e-5 and err(dps, b.grad) < 1e-5
    # AdamW vs torch.optim.AdamW, 3 steps, odd length
    n = 1003
    p0 = rnd(n, seed=7)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-2)
    pf = torch.zeros(1008, device=DEV)
    pf[:n] = p0
    m, v = torch.zeros(1008, device=DEV), torch.zeros(1008, device=DEV)
    for step in range(1, 4):
        g = rnd(n, seed=10 + step)
        p.grad = g.clone()
        opt.step()
        gf = torch.zeros(1008, device=DEV)
        gf[:n] = g
        o.adamw(pf, gf, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step)
    assert err(pf[:n], p.detach()) < 1e-6
    # embed rows / reduce rows / act bwd
    tab = rnd(7, 64, seed=20)
    idx = torch.tensor([3, 3, 0, 6, 1], dtype=torch.int32, device=DEV)
    outr = torch.empty(5, 64, device=DEV)
    o.embed_rows(tab, idx, outr, 5, 64)
    assert torch.equal(outr, tab[idx.long()])
    dtab = torch.empty(7, 64, device=DEV)
    o.embed_rows_bwd(outr.data_ptr(), 64, idx, dtab, 5, 7, 64)
    assert err(dtab, torch.zeros(7, 64, device=DEV, dtype=torch.float64).index_add_(0, idx.long(), outr.double())) < 1e-6
    S, Bb, H = 5, 4, 32
    src = rnd(S * Bb, H, seed=21)
    d1 = torch.empty(S, H, device=DEV)
    o.reduce_rows(src.data_ptr(), H, d1.data_ptr(), H, S, Bb, Bb, 1, H)
    assert err(d1, src.double().reshape(S, Bb, H).sum(1)) < 1e-6
    d2 = torch.empty(Bb, H, device=DEV)
    o.reduce_rows(src.data_ptr(), H, d2.data_ptr(), H, Bb, S, 1, Bb, H)
    assert err(d2, src.double().reshape(S, Bb, H).sum(0)) < 1e-6
    yv, dyv = rnd(100, 32, seed=22), rnd(100, 32, seed=23)
    oo = torch.empty(100, 32, device=DEV)
    o.act_bwd(dyv, yv, 0.01, oo)
    assert err(oo, torch.where(yv > 0, dyv, 0.01 * dyv)) < 1e-7


@pytest.mark.parametrize("B,seed", [(1, 0), (7, 1), (64, 2)])
def test_csr_build_matches_host(B, seed):
    """dosx_csr_build (device) == batch._build_meta_host (numpy) on shuffled, PyG-style index tensors."""
    import numpy as np
    from dostransformer_amd import synth
    from dostransformer_amd.batch import _build_meta_host
    o = ops()
    g = synth.phonon_batch(B, seed=seed, dtype=torch.float32, sort_edges=False)
    ei = g.edge_index.clone()
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(seed))
    ei = ei[:, perm]                                         # arbitrary edge order
    ref = _build_meta_host(ei.numpy(), g.batch.numpy(), B, None, presorted=False)
    r = o.csr_build(ei.to(DEV), g.batch.to(DEV), B)
    torch.cuda.synchronize()
    for k in ("src", "dst", "rowptr_dst", "perm_src", "rowptr_src", "graph_ptr", "node_graph", "dense_row"):
        assert torch.equal(r[k].cpu(), getattr(ref, k)), k
    assert torch.equal(r["edge_perm"].cpu(), ref.edge_perm)
    assert torch.equal(r["inv_deg"].cpu(), ref.inv_deg)
    assert int(r["n_max"].item()) == ref.n_max


def test_wgrad_grouped_is_bitwise_the_single_launches():
    """dosx_wgrad_grouped over a mixed job list (all four fast prologues, a gather, an unaligned job that falls back to
    its own launch, more than 12 jobs so that two grouped launches happen) == the same jobs through dosx_wgrad."""
    o = ops()
    jobs, outs = [], []

    def add(M, N, K, seed, **kw):
        dy, a = rnd(M, N, seed=seed), rnd(M, K, seed=seed + 1)
        ns = o.wgrad_splits(M, N, K)
        pair = []
        for _ in range(2):
            pair.append((torch.full((ns, N, K), float("nan"), device=DEV), torch.full((ns, N), float("nan"), device=DEV)))
        segs = kw.pop("segs", None) or [o.seg(a)]
        keep = (dy, a, segs)
        jobs.append((M, N, dy, segs, pair, ns, kw, keep))

    H = 64
    add(900, 128, 64, 1)
    add(3000, 256, 128, 3, pro=o.PRO_PRELU, pro_alpha=torch.tensor([0.25], device=DEV))
    add(1000, 64, 128, 5, pro=o.PRO_LN_PRELU, pro_gamma=rnd(128, seed=50), pro_beta=rnd(128, seed=51),
        pro_alpha=torch.tensor([0.1], device=DEV))
    add(2000, 256, 64, 7, pro=o.PRO_ROWLN, pro_gamma=rnd(64, seed=52), pro_beta=rnd(64, seed=53),
        pro_stats=torch.rand(2000, 2, device=DEV))
    add(333, 64, 118, 9)                                   # K % 4 != 0: not groupable, own launch
    x = rnd(50, H, seed=60)
    idx = torch.randint(0, 50, (1200,), device=DEV, dtype=torch.int32)
    e = rnd(1200, H, seed=61)
    add(1200, 128, 2 * H, 11, segs=[o.seg(x, rmap=o.rowmap(idx=idx)), o.seg(e)])
    for k in range(9):
        add(500 + 100 * k, 64, 64, 20 + 2 * k)
    descs = []
    for M, N, dy, segs, pair, ns, kw, _ in jobs:
        o.wgrad(M, N, o.seg(dy), segs, pair[0][0], pair[0][1], ns, **kw)
        descs.append(o.wgrad_desc(M, N, o.seg(dy), segs, pair[1][0], pair[1][1], ns, **kw))
    o.wgrad_grouped(descs)
    torch.cuda.synchronize()
    for k, (M, N, dy, segs, pair, ns, kw, _) in enumerate(jobs):
        assert torch.equal(pair[0][0], pair[1][0]) and torch.equal(pair[0][1], pair[1][1]), k
        assert not torch.isnan(pair[1][0]).any()


@pytest.mark.parametrize("M,H", [(7, 16), (100, 128), (3000, 256)])
def test_rownorm_bwd_act(M, H):
    """dosx_rownorm_bwd_act == autograd of  xhat = LN_noaffine(leaky_relu(pre))  plus an extra gradient on leaky_relu(pre)."""
    o = ops()
    pre = rnd(M, H, seed=1).double().requires_grad_(True)
    y = F.leaky_relu(pre, 0.01)
    xhat = F.layer_norm(y, (H,), None, None, 1e-5)
    dxhat, dy_extra = rnd(M, H, seed=2), rnd(M, H, seed=3)
    (xhat * dxhat.double()).sum().backward(retain_graph=True)
    y.backward(dy_extra.double())
    yf = y.detach().float()
    rstd = (1 / torch.sqrt(yf.double().var(1, unbiased=False) + 1e-5)).float()
    out = torch.empty(M, H, device=DEV)
    o.rownorm_bwd_act(dxhat, xhat.detach().float().contiguous(), rstd, dy_extra, yf, 0.01, out, M, H)
    assert err(out, pre.grad) < 5e-5


@pytest.mark.parametrize("M,H", [(33, 32), (100, 64), (3264, 128), (6528, 128), (1, 128)])
def test_ffn_fused_forward(M, H):
    """dosx_ffn_fwd == the two GEMMs of layers/transformer.py:141-148 (pre-norm FFN with residual)."""
    o = ops()
    assert o.ffn_supported(H)
    x = rnd(M, H, seed=1)
    g, b = rnd(H, seed=2), rnd(H, seed=3)
    w1, b1 = rnd(4 * H, H, seed=4, scale=0.2), rnd(4 * H, seed=5)
    w2, b2 = rnd(H, 4 * H, seed=6, scale=0.2), rnd(H, seed=7)
    mu = x.mean(1, keepdim=True)
    rstd = 1 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5)
    stats = torch.cat([mu, rstd], 1).contiguous()
    h = torch.empty(M, 4 * H, device=DEV)
    out = torch.empty(M, H, device=DEV)
    o.ffn_fwd(M, H, x, stats, g, b, w1, b1, w2, b2, h, out)
    xd = x.double()
    ln = F.layer_norm(xd, (H,), g.double(), b.double(), 1e-5)
    href = torch.relu(ln @ w1.double().T + b1.double())
    assert err(h, href) < TOL
    ref = xd + href @ w2.double().T + b2.double()
    assert err(out, ref) < TOL
    # same launch with the encoder's final LayerNorm fused into the row epilogue
    fg, fb = rnd(H, seed=8), rnd(H, seed=9)
    xhat, rstd_o, y = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, H, device=DEV)
    o.ffn_fwd(M, H, x, stats, g, b, w1, b1, w2, b2, h, y, fin=(fg, fb, xhat, rstd_o))
    mu2 = ref.mean(1, keepdim=True)
    rs2 = 1 / torch.sqrt(ref.var(1, unbiased=False, keepdim=True) + 1e-5)
    assert err(xhat, (ref - mu2) * rs2) < 5e-5 and err(rstd_o, rs2[:, 0]) < 5e-5
    assert err(y, (ref - mu2) * rs2 * fg.double() + fb.double()) < 5e-5


@pytest.mark.parametrize("M,H", [(33, 32), (100, 64), (3264, 128), (6528, 128), (1, 128), (50, 96)])
def test_ffn_fused_backward(M, H):
    """dosx_ffn_bwd == autograd of the pre-norm FFN half layer: dh (masked), dx (incl. the residual path), LN1 dgamma/dbeta."""
    o = ops()
    x = rnd(M, H, seed=1).double().requires_grad_(True)
    g = rnd(H, seed=2).double().requires_grad_(True)
    b = rnd(H, seed=3).double().requires_grad_(True)
    w1, b1 = rnd(4 * H, H, seed=4, scale=0.2), rnd(4 * H, seed=5)
    w2, b2 = rnd(H, 4 * H, seed=6, scale=0.2), rnd(H, seed=7)
    dy = rnd(M, H, seed=8)
    ln = F.layer_norm(x, (H,), g, b, 1e-5)
    pre = ln @ w1.double().T + b1.double()
    pre.retain_grad()
    href = torch.relu(pre)
    out = x + href @ w2.double().T + b2.double()
    out.backward(dy.double())
    xf = x.detach().float()
    mu = xf.mean(1, keepdim=True)
    rstd = 1 / torch.sqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-5)
    stats = torch.cat([mu, rstd], 1).contiguous()
    rows = o.ffn_bwd_partial_rows(M)
    dh = torch.empty(M, 4 * H, device=DEV)
    dx = torch.empty(M, H, device=DEV)
    part = torch.full((rows, 2 * H), float("nan"), device=DEV)
    o.ffn_bwd(M, H, dy, href.detach().float().contiguous(), xf, stats, g.detach().float(), w1, w2, dh, dx, part)
    assert err(dh, pre.grad) < TOL
    assert err(dx, x.grad) < 5e-5
    ps = part.double().sum(0)
    assert err(ps[:H], g.grad) < 5e-5 and err(ps[H:], b.grad) < 5e-5


# ---- §8f-3 periodic neighbour list (dosx_neighbor_count / _fill) --------------------------------------------------
def _random_crystals(seed, sizes):
    rng = np.random.default_rng(seed)
    pos, cells = [], []
    for n in sizes:
        cell = np.diag(rng.uniform(2.5, 6.0, 3)) + rng.uniform(-1.2, 1.2, (3, 3))       # triclinic, well conditioned
        frac = rng.uniform(-1.5, 2.5, (n, 3))                                           # NOT wrapped into the cell
        pos.append(frac @ cell)
        cells.append(cell)
    return pos, cells


@pytest.mark.parametrize("cutoff,si", [(3.0, True), (5.0, True), (4.0, False)])
def test_neighbor_list_matches_oracle(cutoff, si):
    """Same edges, same order (crystal, i, j, shift), bit-identical edge_vec as the brute-force restatement."""
    from oracle.dos_oracle import neighbor_list_bruteforce
    sizes = [1, 2, 12, 5, 30, 1, 7]
    pos, cells = _random_crystals(7, sizes)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    out = ops().neighbor_list(torch.from_numpy(np.concatenate(pos)).to(DEV), torch.from_numpy(np.stack(cells)).to(DEV),
                            torch.from_numpy(ptr).to(DEV), cutoff, self_interaction=si)
    eptr = out["edge_ptr"].cpu().numpy()
    assert eptr[-1] == out["src"].numel() > 0
    for c, (p, cell) in enumerate(zip(pos, cells)):
        i, j, S, D = neighbor_list_bruteforce(p, cell, cutoff, si)
        a, b = eptr[c], eptr[c + 1]
        assert b - a == len(i), c
        assert (out["crystal"][a:b].cpu().numpy() == c).all()
        assert (out["src"][a:b].cpu().numpy() == i).all() and (out["dst"][a:b].cpu().numpy() == j).all()
        assert (out["shift"][a:b].cpu().numpy() == S).all()
        assert np.array_equal(out["edge_vec"][a:b].cpu().numpy(), D)


def test_neighbor_list_known_answers_and_slab():
    """fcc coordination shells (12, 6, 24) straight from the kernel; a non-periodic axis only keeps shift 0 there."""
    fcc = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], float)
    cell = np.eye(3)[None]
    ptr = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    for rc, expect in [(0.71, 12), (1.01, 18), (1.23, 42)]:
        out = ops().neighbor_list(torch.from_numpy(fcc).to(DEV), torch.from_numpy(cell).to(DEV), ptr, rc, self_interaction=False)
        assert (np.bincount(out["src"].cpu().numpy(), minlength=4) == expect).all()
    slab = ops().neighbor_list(torch.from_numpy(fcc).to(DEV), torch.from_numpy(cell).to(DEV), ptr, 1.01,
                             self_interaction=False, pbc=(True, True, False))
    assert int(slab["shift"][:, 2].abs().max()) == 0 and 0 < slab["src"].numel() < 4 * 18


def test_build_data_all_feeds_the_model():
    """structures -> featurize.build_data_all -> collate -> model: schema of `utils.py:291-301`, x = mass one-hot."""
    from dostransformer_amd import featurize
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from oracle.dos_oracle import neighbor_list_bruteforce
    rng = np.random.default_rng(3)
    pos, cells = _random_crystals(11, [2, 5, 3])
    entries = [{"symbols": [featurize.SYMBOLS[k] for k in rng.integers(0, 90, p.shape[0])], "positions": p, "cell": c,
                "phdos": rng.uniform(0, 1, 51), "crystal_system": cs, "mp_id": f"mp-{k}"}
               for k, (p, c, cs) in enumerate(zip(pos, cells, ["Cubic", "Monoclinic", "Triclinic"]))]
    data = featurize.build_data_all(entries, r_max=4.0, device=DEV, dtype=torch.float32)
    assert [int(d["system"]) for d in data] == [0, 5, 6]
    for d, e in zip(data, entries):
        i, j, S, D = neighbor_list_bruteforce(e["positions"], e["cell"], 4.0, True)
        assert d["edge_index"].shape == (2, len(i)) and np.array_equal(d["edge_index"].numpy(), np.stack([i, j]))
        assert np.allclose(d["edge_vec"].numpy(), D, atol=1e-6)
        z = [featurize.SYMBOLS.index(s) for s in e["symbols"]]
        assert d["x"].shape == (len(z), 118) and (d["x"] != 0).sum() == len(z)
        assert np.allclose(d["x"][range(len(z)), z].numpy(), [featurize.ATOMIC_MASSES[k] for k in z], rtol=1e-6)
        assert d["phdos"].shape == (1, 51)
    torch.manual_seed(0)
    model = DOSTransformer_phonon(2, 1, 118, 4, 32, DEV, 0.0).to(DEV)
    g = collate(data).to(DEV)
    with torch.no_grad():
        dg, x, ds = model(g)
    assert dg.shape == (3, 51) and torch.isfinite(dg).all() and torch.isfinite(ds).all()
