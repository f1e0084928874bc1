"""dosx_gemm / dosx_gemm_pair / the weight-gradient kernels and their in-launch reductions (csrc/gemm.hip) against float64 and
against each other's forms - every `nn.Linear` of the path with the LayerNorm / PReLU / ReLU work fused around it."""
import copy
import ctypes as C
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import rmse  # noqa: F401
from tests.gpu_util import (DEV, TOL, _FakeDist, _Hog, _attn_ref, _descs, _fat_crystals, _fatten, _graph, _mixed_jobs, _node_block, _philox_mask_numpy, _phonon, _random_crystals, _reduce, _ref, _scratch, _sliver_case, err, ops, prelu, rnd)  # noqa: F401

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (128, 128, 32), (5000, 1024, 256), (4111, 256, 1024), (12864, 1024, 256)])
@pytest.mark.parametrize("wl", [0, 1])
def test_gemm_bf16x3_meets_the_fp32_tolerance(M, N, K, wl):
    """The split-bf16 GEMM (csrc/gemm_bf16x3.hip; a measurement next to dosx_gemm): fp32 operands split into three
    bf16 terms on the fly, six products on the bf16 matrix pipe, fp32 accumulation - held to dosx_gemm's OWN tolerance against
    float64 (TOL, unchanged) and to a small multiple of the exact-fp32 kernel's error on the same operands; ragged M, both weight
    layouts, bias."""
    o = ops()
    a = rnd(M, K, seed=1)
    w = rnd(N, K, seed=2) if wl == 0 else rnd(K, N, seed=2)
    b = rnd(N, seed=3)
    assert o.gemm_bf16x3_supported(M, N, K)
    out = torch.empty(M, N, device=DEV)
    o.gemm_bf16x3(a, w, out, bias=b, w_layout=wl)
    ref = a.double() @ (w.double().T if wl == 0 else w.double()) + b.double()
    e3 = err(out, ref)
    out32 = torch.empty(M, N, device=DEV)
    o.gemm(M, N, [o.seg(a)], w, out32, w_layout=wl, bias=b)
    e32 = err(out32, ref)
    assert e3 < TOL, (e3, e32)
    assert e3 <= 4.0 * e32 + 1e-7, (e3, e32)
    # A = I with an asymmetric W: no transposed tile, and the three terms of every weight element add up to it
    if M == 300:
        eye = torch.zeros(M, K, device=DEV)
        eye[:K] = torch.eye(K, device=DEV)
        wa = (torch.arange(N * K, device=DEV, dtype=torch.float32).reshape(w.shape) % 97) / 7.0
        o.gemm_bf16x3(eye, wa, out, w_layout=wl)
        want = wa.T if wl == 0 else wa
        assert err(out[:K], want.contiguous()) < 2e-7 and float(out[K:].abs().max()) == 0.0


@pytest.mark.parametrize("wl", [0, 1])
def test_gemm_bf16x3_epilogues(wl):
    """relu, residual and the ReLU-mask epilogue (what the feed-forward half of an encoder layer asks of its GEMMs) against float64
    and against dosx_gemm's own epilogues."""
    o = ops()
    M, N, K = 777, 256, 128
    a, b = rnd(M, K, seed=1), rnd(N, seed=3)
    w = rnd(N, K, seed=2) if wl == 0 else rnd(K, N, seed=2)
    res, h = rnd(M, N, seed=4), rnd(M, N, seed=5)
    z = a.double() @ (w.double().T if wl == 0 else w.double())
    out = torch.empty(M, N, device=DEV)
    o.gemm_bf16x3(a, w, out, bias=b, w_layout=wl, act=o.ACT_RELU)
    assert err(out, torch.relu(z + b.double())) < TOL
    o.gemm_bf16x3(a, w, out, bias=b, w_layout=wl, res=res)
    assert err(out, z + b.double() + res.double()) < TOL
    o.gemm_bf16x3(a, w, out, w_layout=wl, mask=h)
    assert err(out, z * (h.double() > 0)) < TOL
    if wl == 1:                                         # (dosx_gemm has the ReLU-mask epilogue for input gradients only)
        out32 = torch.empty(M, N, device=DEV)
        o.gemm(M, N, [o.seg(a)], w, out32, w_layout=wl, epi=o.EPI_RELU_MASK, aux=h)
        assert torch.equal(out == 0, out32 == 0)        # the same gate, element for element


def test_gemm_bf16x3_rejects_shapes_it_does_not_tile():
    from dostransformer_amd._lib import DosxError
    o = ops()
    a, w, out = rnd(64, 40, seed=1), rnd(128, 40, seed=2), torch.empty(64, 128, device=DEV)
    assert not o.gemm_bf16x3_supported(64, 128, 40) and not o.gemm_bf16x3_supported(64, 100, 64)
    with pytest.raises(DosxError):
        o.gemm_bf16x3(a, w, out)


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


@pytest.mark.parametrize("M,N,K", [(1000, 256, 384), (333, 128, 118), (70, 64, 64), (5000, 128, 512), (40, 16, 24),
                                    (9344, 256, 384), (6528, 512, 128), (456, 256, 256), (64, 128, 320), (3, 8, 4),
                                    # M >= 16384 and N >= 128: 128 x 64 tiles (two sub-tiles per matrix wave); ragged N and K
                                    # (eDOS-size jobs)
                                    (20000, 256, 384), (16500, 192, 128), (16400, 128, 64), (17000, 328, 72),
                                    (25728, 1024, 256), (17880, 512, 768), (16390, 136, 200)])
def test_wgrad_finished_mode(M, N, K):
    """dW = dY^T A and db = sum_m dY written by the weight-gradient kernel itself (no reduce_partials launch), against
    float64; a second launch on the same counters (they must be back at zero) gives the same bits; accumulate adds."""
    o = ops()
    dy, a = rnd(M, N, seed=1), rnd(M, K, seed=2)
    ns = o.wgrad_splits(M, N, K)
    slab, slab_b = _scratch(o, N, K, ns)
    dw = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    g = o.wgrad_desc(M, N, o.seg(dy), [o.seg(a)], slab, slab_b, ns, dst=dw, dst_bias=db)
    o.wgrad_grouped([g])
    torch.cuda.synchronize()
    ref_w, ref_b = dy.double().T @ a.double(), dy.double().sum(0)
    assert err(dw, ref_w) < TOL and err(db, ref_b) < TOL
    first_w, first_b = dw.clone(), db.clone()
    dw.fill_(float("nan"))
    o.wgrad_grouped([g])                       # same descriptor, same counters
    torch.cuda.synchronize()
    assert torch.equal(dw, first_w) and torch.equal(db, first_b)
    g.accumulate = 1
    o.wgrad_grouped([g])
    torch.cuda.synchronize()
    assert err(dw, 2 * ref_w) < TOL and err(db, 2 * ref_b) < TOL


def test_wgrad_finished_grouped_is_bitwise_the_single_launches_and_reproducible():
    """The same jobs (all prologues, a gather, an unaligned K, more than 8 jobs -> several grouped launches) through
    dosx_wgrad one by one and through dosx_grad_flush: bitwise equal (the summation order over the M-splits is fixed, whoever
    arrives last); ten more grouped launches, with the scratch slabs re-read by another kernel in between so that stale
    lines sit in the caches, reproduce the same bits."""
    o = ops()
    jobs = _mixed_jobs(o)
    d1, o1 = _descs(o, jobs)
    for g in d1:
        o._call("dosx_wgrad", __import__("ctypes").byref(g), o._stream())
    d2, o2 = _descs(o, jobs)
    o.wgrad_grouped(d2)
    torch.cuda.synchronize()
    for k, ((w1, b1, _, _), (w2, b2, _, _)) in enumerate(zip(o1, o2)):
        assert not torch.isnan(w2).any(), k
        assert torch.equal(w1, w2), k
        if b1 is not None:
            assert torch.equal(b1, b2), k
    for j, (w, b, _, _) in zip(jobs, o2):
        if "pro" not in j["kw"] and len(j["segs"]) == 1:
            assert err(w, j["dy"].double().T @ j["a"].double()) < TOL
    ref = [(w.clone(), None if b is None else b.clone()) for w, b, _, _ in o2]
    junk = torch.zeros((), device=DEV)
    for it in range(10):
        for w, b, slab, slab_b in o2:
            if slab is not None:
                junk += torch.nan_to_num(slab).sum()          # plain loads of the scratch lines: they stay in L1 / L2
            w.fill_(float("nan"))
        o.wgrad_grouped(d2)
        torch.cuda.synchronize()
        for k, ((w, b, _, _), (rw, rb)) in enumerate(zip(o2, ref)):
            assert torch.equal(w, rw), (it, k)
            if b is not None:
                assert torch.equal(b, rb), (it, k)


def test_grad_flush_carries_the_row_partial_reductions():
    """dosx_grad_flush: weight-gradient jobs + row-partial reductions in one grid == dosx_reduce_partials on the same jobs
    (bitwise: the reduction body is shared), for ragged counts, odd slice numbers, more than 40 jobs and no wgrad jobs."""
    o = ops()
    g = torch.Generator().manual_seed(3)
    rjobs, refs, outs, keep = [], [], [], []
    shapes = [(204, 256, 256), (195, 513, 512), (7, 1, 1), (64, 130, 100), (33, 1024, 1024), (1, 8, 8)] + [(17, 64 + 4 * k, 60 + 4 * k) for k in range(45)]
    for rows, stride, count in shapes:
        src = torch.randn(rows, stride, generator=g).to(DEV)
        d1, d2 = torch.full((count,), float("nan"), device=DEV), torch.full((count,), float("nan"), device=DEV)
        keep.append(src)
        rjobs.append((src.data_ptr(), d1.data_ptr(), rows, stride, count, 0))
        refs.append((src.data_ptr(), d2.data_ptr(), rows, stride, count, 0))
        outs.append((d1, d2, src, count))
    jobs = _mixed_jobs(o)[:3]
    descs, wout = _descs(o, jobs)
    o.grad_flush(descs, rjobs)
    sink = o.GradSink(DEV)
    sink._reduce(refs)
    torch.cuda.synchronize()
    for k, (d1, d2, src, count) in enumerate(outs):
        assert torch.equal(d1, d2), k
        assert err(d1, src.double()[:, :count].sum(0)) < TOL
    for w, b, _, _ in wout:
        assert not torch.isnan(w).any()
    # reductions only
    for d1, _, _, _ in outs:
        d1.fill_(float("nan"))
    o.grad_flush([], rjobs)
    torch.cuda.synchronize()
    for k, (d1, d2, _, _) in enumerate(outs):
        assert torch.equal(d1, d2), k


def test_sink_serialises_jobs_that_share_a_gradient():
    """Two weight-gradient jobs into the same parameter gradient (a weight used twice) in one flush: the second one runs in a
    later launch and accumulates."""
    o = ops()
    from dostransformer_amd import functional as Fn
    M, N, K = 700, 64, 128
    dy1, a1, dy2, a2 = rnd(M, N, seed=1), rnd(M, K, seed=2), rnd(M, N, seed=3), rnd(M, K, seed=4)
    G = {"w": torch.full((N, K), float("nan"), device=DEV), "b": torch.full((N,), float("nan"), device=DEV)}
    sink = o.GradSink(DEV)
    Fn._wgrad_linear(sink, G, "w", "b", M, N, o.seg(dy1), [o.seg(a1)], keep=(dy1,))
    Fn._wgrad_linear(sink, G, "w", "b", M, N, o.seg(dy2), [o.seg(a2)], keep=(dy2,))
    sink.flush()
    torch.cuda.synchronize()
    assert err(G["w"], dy1.double().T @ a1.double() + dy2.double().T @ a2.double()) < TOL
    assert err(G["b"], dy1.double().sum(0) + dy2.double().sum(0)) < TOL


@pytest.mark.parametrize("B,S", [(64, 51), (32, 7), (8, 51), (64, 201)])
def test_wgrad_blocked_row_maps(B, S):
    """The heads' weight gradients (`DOSTransformer_phonon.py:93-95,105-109`): dY = the rows of ONE prediction branch inside a
    [S*2B, H] tensor (div/mod row map), A = cat[energies broadcast over the crystals, graph broadcast over the bins] - through
    the buffer-addressed fast path when the maps are chunk-aligned (32 | B) and through the generic path otherwise; finished
    mode, grouped, against float64."""
    o = ops()
    H = 64
    M = S * B
    dpre = rnd(S * 2 * B, H, seed=1)
    en, gr = rnd(S, H, seed=2), rnd(B, H, seed=3)
    for branch in (0, 1):
        dmap = o.rowmap(d=B, m=2 * B, c=1, off=branch * B)
        segs = [o.seg(en, rmap=o.rowmap(d=B, m=1, c=0)), o.seg(gr, rmap=o.rowmap(d=B, m=0, c=1))]
        ns = o.wgrad_splits(M, H, 2 * H)
        slab, slab_b = _scratch(o, H, 2 * H, ns)
        dw = torch.full((H, 2 * H), float("nan"), device=DEV)
        db = torch.full((H,), float("nan"), device=DEV)
        g = o.wgrad_desc(M, H, o.seg(dpre, rmap=dmap), segs, slab, slab_b, ns, dst=dw, dst_bias=db)
        o.wgrad_grouped([g])
        torch.cuda.synchronize()
        dy = dpre.double().reshape(S, 2, B, H)[:, branch].reshape(M, H)
        a = torch.cat([en.double()[:, None, :].expand(S, B, H), gr.double()[None, :, :].expand(S, B, H)], 2).reshape(M, 2 * H)
        assert err(dw, dy.T @ a) < TOL and err(db, dy.sum(0)) < TOL


def test_randomised_sweep_of_the_in_launch_reductions():
    """Seeded random shapes through the two in-launch reductions of round 3: finished-mode weight gradients (any M, N, K incl.
    unaligned K, with / without bias) and the message GEMM's segment sums over graphs with random in-degrees (isolated
    nodes, over-full nodes of up to 400 edges, exact multiples of the tile height) - against float64 / the stand-alone
    segment reduction."""
    from dostransformer_amd import functional as Fn
    from dostransformer_amd.batch import seg_tiles_host
    o = ops()
    rng = np.random.default_rng(7)
    for trial in range(24):
        M = int(rng.integers(1, 12000))
        N = int(rng.choice([4, 16, 64, 100, 128, 256, 384]))
        K = int(rng.choice([4, 24, 41, 64, 118, 128, 200, 256, 512]))
        dy, a = rnd(M, N, seed=100 + trial), rnd(M, K, seed=200 + trial)
        ns = o.wgrad_splits(M, N, K)
        bias = bool(trial % 2)
        slab, slab_b = _scratch(o, N, K, ns, bias)
        dw = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV) if bias else None
        o.wgrad_grouped([o.wgrad_desc(M, N, o.seg(dy), [o.seg(a)], slab, slab_b, ns, dst=dw, dst_bias=db)])
        torch.cuda.synchronize()
        assert err(dw, dy.double().T @ a.double()) < TOL, (M, N, K)
        if bias:
            assert err(db, dy.double().sum(0)) < TOL, (M, N, K)
    for trial in range(12):
        n = int(rng.integers(1, 60))
        deg = rng.integers(0, 30, size=n)
        deg[rng.random(n) < 0.2] = 0
        for _ in range(int(rng.integers(0, 4))):
            deg[int(rng.integers(0, n))] = int(rng.choice([48, 49, 96, 97, 144, 200, 400]))
        E = int(deg.sum())
        if E == 0:
            continue
        H = int(rng.choice([16, 64, 128, 256]))
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        dst = torch.from_numpy(np.repeat(np.arange(n), deg).astype(np.int32)).to(DEV)
        src = torch.from_numpy(rng.integers(0, n, size=E).astype(np.int32)).to(DEV)
        tiles = torch.from_numpy(seg_tiles_host(rowptr)).to(DEV)
        rp = torch.from_numpy(rowptr.astype(np.int32)).to(DEV)
        inv = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(DEV)
        gen = torch.Generator().manual_seed(trial)
        P = {"k.0.weight": torch.randn(2 * H, 3 * H, generator=gen) * 0.1, "k.0.bias": torch.randn(2 * H, generator=gen),
             "k.1.weight": torch.randn(2 * H, generator=gen), "k.1.bias": torch.randn(2 * H, generator=gen),
             "k.2.weight": torch.tensor([0.25]), "k.3.weight": torch.randn(H, 2 * H, generator=gen) * 0.1,
             "k.3.bias": torch.randn(H, generator=gen)}
        P = {k: v.to(DEV) for k, v in P.items()}
        x, e = torch.randn(n, H, generator=gen).to(DEV), torch.randn(E, H, generator=gen).to(DEV)
        a = Fn.SegList([o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(e)], [x, e])
        scale = inv if trial % 2 else None
        msg, _ = Fn.mlp_ln_fwd(P, "k", a, E, H)
        agg0, e0 = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV)
        o.segment_reduce(msg, rp, scale, agg0, e, e0, n, E, H)
        agg1, e1 = torch.full((n, H), float("nan"), device=DEV), torch.full((E, H), float("nan"), device=DEV)
        Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(tiles, rp, scale, agg1, e, e1))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(agg1).all()), (trial, deg.tolist())
        assert float((agg1 - agg0).abs().max()) <= 4e-6 * float(agg0.abs().max() + 1e-6), (trial, deg.tolist())
        assert torch.equal(e1, e0)


@pytest.mark.parametrize("M,N,K,mapped", [(6528, 128, 256, True), (300, 256, 128, False), (4000, 64, 96, True), (33, 512, 64, False)])
def test_gemm_writes_normalised_rows_too(M, N, K, mapped):
    """DosxGemm.norm_out: the plain epilogue also writes LayerNorm(out) without affine and its rstd at the OUTPUT rows (through
    out_map) - what dosx_rownorm on the finished output gives (the heads' GEMMs feed the self-attention encoder's stale keys
    this way, DOSTransformer_phonon.py:90-97)."""
    o = ops()
    a, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3)
    rows = 2 * M if mapped else M
    d = 64 if mapped else 1
    omap = o.rowmap(d=d, m=2 * d, c=1, off=d) if mapped else None       # blocks of d rows into every other block of 2d
    if mapped and M % d:
        pytest.skip("mapped case wants M % 64 == 0")
    out = torch.full((rows, N), float("nan"), device=DEV)
    nrm = torch.full((rows, N), float("nan"), device=DEV)
    rstd = torch.full((rows,), float("nan"), device=DEV)
    o.gemm(M, N, [o.seg(a)], w, out, bias=bias, act=o.ACT_LEAKY, act_slope=0.01, out_map=omap, norm_out=nrm, norm_rstd=rstd)
    torch.cuda.synchronize()
    y = torch.nn.functional.leaky_relu(a.double() @ w.double().T + bias.double(), 0.01)
    idx = torch.arange(M, device=DEV)
    if mapped:
        idx = (idx // d) * 2 * d + idx % d + d
    assert err(out[idx], y) < 2e-5
    mu, var = y.mean(1, keepdim=True), y.var(1, unbiased=False, keepdim=True)
    assert err(nrm[idx], (y - mu) / torch.sqrt(var + 1e-5)) < 5e-5
    assert err(rstd[idx], 1 / torch.sqrt(var[:, 0] + 1e-5)) < 5e-5
    other = torch.ones(rows, dtype=torch.bool, device=DEV)
    other[idx] = False
    assert bool(torch.isnan(nrm[other]).all()) and bool(torch.isnan(rstd[other]).all())      # nothing else touched


def test_stress_weight_gradient_tickets_under_a_bandwidth_hog():
    """VERDICT r3 item 5(a), weight gradients: 2 000 back-to-back grouped finished-mode launches (3 jobs each, their
    M-splits reduced by the last arriving workgroup of every tile - write-through publish + ticket, csrc/gemm.hip
    wgrad_finish) while a second stream streams 1 GiB per copy through HBM.  EVERY launch is compared on the device, bit for
    bit, with the result of the same jobs through the single-job launches; the destinations are NaN-filled in between."""
    o = ops()
    shapes = [(3000, 256, 128), (9344, 256, 384), (6528, 128, 512)]
    descs, dsts, refs = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, a = rnd(M, N, seed=10 + i), rnd(M, K, seed=20 + i)
        ns = o.wgrad_splits(M, N, K)
        assert ns > 1
        slab, slab_b = _scratch(o, N, K, ns)
        dw, db = torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV)
        g = o.wgrad_desc(M, N, o.seg(dy), [o.seg(a)], slab, slab_b, ns, dst=dw, dst_bias=db)
        o._call("dosx_wgrad", C.byref(g), o._stream())           # the single-job launch: the reference bits
        torch.cuda.synchronize()
        assert err(dw, dy.double().T @ a.double()) < TOL
        refs.append((dw.clone(), db.clone()))
        descs.append(g)
        dsts.append((dw, db, dy, a, slab, slab_b))
    hog = _Hog()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    n_launch = 2000
    for it in range(n_launch):
        if it % 4 == 0:
            hog.feed()
        for dw, db, *_ in dsts:
            dw.fill_(float("nan"))
            db.fill_(float("nan"))
        o.wgrad_grouped(descs)
        for (dw, db, *_), (rw, rb) in zip(dsts, refs):
            bad += (dw != rw).sum() + (db != rb).sum()          # NaN != x counts too
    torch.cuda.synchronize()
    assert hog.n >= n_launch // 4 and int(bad) == 0, int(bad)


@pytest.mark.parametrize("M,N,K,mapped,res", [(6528, 128, 128, False, False), (1554, 512, 512, False, True), (450, 256, 256, False, False),
                                              (3264, 128, 256, True, True), (70, 68, 48, False, True), (1, 4, 16, False, False),
                                              (12864, 256, 256, True, False), (25728, 256, 256, False, False)])
def test_sliver_gemm_matches_float64(M, N, K, mapped, res):
    """With dosx_set_sliver_max_gf(2) dosx_gemm routes small plain dgrad GEMMs (w_layout 1, no prologue / bias / activation) to
    the vector-ALU kernel with the co-residable footprint (csrc/gemm.hip: sliver_gemm_kernel): against float64, with a div/mod row map on A
    (the heads' dgrad), a residual, ragged tiles; larger problems keep the MFMA kernels."""
    from dostransformer_amd import _lib
    from dostransformer_amd._lib import Gemm
    o = ops()
    _lib.load().dosx_set_sliver_max_gf(2.0)             # (an experiment switch: off by default)
    try:
        _sliver_case(o, _lib, Gemm, M, N, K, mapped, res)
    finally:
        _lib.load().dosx_set_sliver_max_gf(0.0)


@pytest.mark.parametrize("M1,M2,N,K1,K2", [(3264, 3264, 128, 256, 320), (51, 51, 128, 256, 320), (12864, 12864, 256, 512, 640),
                                            (700, 1900, 64, 128, 160), (3264, 3264, 128, 256, 118)])
def test_gemm_pair_is_the_two_gemms(M1, M2, N, K1, K2):
    """dosx_gemm_pair: two problems (the output heads `fc` / `fc_prompt`, DOSTransformer_phonon.py:93-109: mod-B gathered segments,
    LeakyReLU, remapped output rows, normalised copy) in one grid against the two separate launches - same tile arithmetic per
    row when the tile height agrees, rounding otherwise; the last case (unaligned K) falls back to two launches."""
    from dostransformer_amd import ops
    torch.manual_seed(3)
    B = 8
    a1, a2 = torch.randn(M1, K1 - 64, device=DEV), torch.randn(M2, K2 - 64, device=DEV)
    gr = torch.randn(B, 64, device=DEV)
    modB = ops.rowmap(d=B, m=0, c=1)
    w1, w2 = torch.randn(N, K1, device=DEV) / K1 ** 0.5, torch.randn(N, K2, device=DEV) / K2 ** 0.5
    b1, b2 = torch.randn(N, device=DEV), torch.randn(N, device=DEV)

    def run(pair):
        out = torch.zeros(M1 + M2, N, device=DEV)
        nrm, rs = torch.zeros(M1 + M2, N, device=DEV), torch.zeros(M1 + M2, device=DEV)
        kw1 = dict(M=M1, N=N, segs=[ops.seg(a1), ops.seg(gr, rmap=modB)], w=w1, out=out, bias=b1, act=ops.ACT_LEAKY, act_slope=0.01,
                   out_map=ops.rowmap(d=1 << 30, m=0, c=1, off=0), norm_out=nrm, norm_rstd=rs)
        kw2 = dict(M=M2, N=N, segs=[ops.seg(a2), ops.seg(gr, rmap=modB)], w=w2, out=out, bias=b2, act=ops.ACT_LEAKY, act_slope=0.01,
                   out_map=ops.rowmap(d=1 << 30, m=0, c=1, off=M1), norm_out=nrm, norm_rstd=rs)
        if pair:
            ops.gemm_pair(kw1, kw2)
        else:
            ops.gemm(**kw1)
            ops.gemm(**kw2)
        return out, nrm, rs
    o2, n2, r2 = run(False)
    o1, n1, r1 = run(True)
    ref1 = torch.cat([a1, gr[torch.arange(M1, device=DEV) % B]], 1).double() @ w1.double().T + b1.double()
    ref2 = torch.cat([a2, gr[torch.arange(M2, device=DEV) % B]], 1).double() @ w2.double().T + b2.double()
    ref = torch.cat([ref1, ref2])
    ref = torch.where(ref >= 0, ref, 0.01 * ref)
    assert err(o1, ref) < 1e-5 and err(o2, ref) < 1e-5
    assert err(o1, o2) < 2e-6 and err(n1, n2) < 2e-5 and err(r1, r2) < 2e-5
    o1b, _, _ = run(True)
    assert torch.equal(o1, o1b)


@pytest.mark.parametrize("M", [16384 + 700, 25728, 16384 + 9000])
def test_mixed_tile_heights_with_the_layernorm_backward_epilogue(M):
    """dosx_gemm's mixed-height grid (gemm_tail_split / gemm_mixed_kernel) on the one epilogue that leaves partial rows: fc1's
    input gradient of a hidden-256 feed-forward layer (transformer.py:141-148 backward) - 64 x 256 tiles for the first 16384
    rows, 32-row tiles (700 tail rows) or 48-row tiles (9344 / 9000 tail rows) behind them, partial-row blocks numbered through
    both parts.  Against torch autograd in fp64; NaN-filled partial buffer: every block is written exactly once."""
    import torch.nn.functional as F
    from dostransformer_amd import ops as o
    H = 256
    g_ = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g_) * scale).to(DEV)
    x = rnd(M, H).double().requires_grad_(True)
    gam = rnd(H).double().requires_grad_(True)
    bet = rnd(H).double().requires_grad_(True)
    w1 = rnd(4 * H, H, scale=0.1)
    dh = rnd(M, 4 * H)
    res = rnd(M, H)
    (F.layer_norm(x, (H,), gam, bet, 1e-5) @ w1.double().T).backward(dh.double())
    mu = x.detach().mean(1)
    rs = 1 / torch.sqrt(x.detach().var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mu, rs], 1).float().contiguous()
    dx = torch.empty(M, H, device=DEV)
    rows = o.gemm_partial_rows(M, H, o.EPI_ROWLN_BWD)
    assert rows == 256 + ((M - 16384 + 47) // 48 if M - 16384 > 8192 else (M - 16384 + 31) // 32)
    part = torch.full((rows, 2 * H), float('nan'), device=DEV)
    o.gemm(M, H, [o.seg(dh)], w1, dx, w_layout=1, epi=o.EPI_ROWLN_BWD, aux=x.detach().float(), aux_stats=stats,
           epi_gamma=gam.detach().float(), res=res, partials=part, partial_ld=2 * H)
    assert err(dx, x.grad + res.double()) < 5e-5
    ps = part.double().sum(0)
    assert not torch.isnan(ps).any()
    assert err(ps[:H], gam.grad) < 5e-5 and err(ps[H:], bet.grad) < 5e-5


@pytest.mark.parametrize("n,N,K", [(400, 256, 128), (37, 128, 64), (700, 512, 256), (20, 256, 128), (330, 64, 32),
                                   # ~17900 / ~8950 edge rows x 512 columns: full rounds of 32-row tiles + a tail of 16-row tiles in
                                   # one grid (gemm_tail_split's 512-column form: the Electron-DOS batch and its 32-crystal shard)
                                   (1450, 512, 256), (725, 512, 256)])
def test_gemm_layernorm_epilogue_with_gathered_addends(n, N, K):
    """DosxGemm.add_p / add_q (round 5): xhat = LN_noaffine(e Wc^T + b + P[src] + Q[dst]) in ONE launch - the EdgeModel's first
    Linear (DOSTransformer_phonon.py:190-197) factored into node products and an edge product of K = H - against float64, at the
    tile shapes the policy picks (48-row / 64-row / 16-row tiles, 128- / 256- / 512-column rows) and P, Q as the two halves of
    one [n, 2N] product (row stride 2N, as the step lays them out)."""
    o = ops()
    src, dst, rp, deg, tiles, E = _graph(n, n + N)
    e, W, b = rnd(E, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    pq = rnd(n, 2 * N, seed=4)
    xhat, rstd = torch.full((E, N), float("nan"), device=DEV), torch.full((E,), float("nan"), device=DEV)
    o.gemm(E, N, [o.seg(e)], W, xhat, bias=b, epi=o.EPI_LN, aux_out=rstd, add_p=pq[:, :N], add_ip=src, add_q=pq[:, N:], add_iq=dst)
    torch.cuda.synchronize()
    z = e.double() @ W.double().T + b.double() + pq[:, :N].double()[src.long()] + pq[:, N:].double()[dst.long()]
    mu, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    r = (var + 1e-5).rsqrt()
    assert err(xhat, (z - mu) * r) < 2e-5 and err(rstd, r[:, 0]) < 2e-5
    # and without the addends the call is what it was
    o.gemm(E, N, [o.seg(e)], W, xhat, bias=b, epi=o.EPI_LN, aux_out=rstd)
    z = e.double() @ W.double().T + b.double()
    mu, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    assert err(xhat, (z - mu) * (var + 1e-5).rsqrt()) < 2e-5


@pytest.mark.parametrize("Mn,Me,H", [(417, 8340, 128), (450, 9000, 128), (17, 300, 128), (417, 8340, 64)])
def test_the_two_encoders_backward_products_as_one_launch(Mn, Me, H):
    """dosx_gemm_pair with two EPI_PRELU_BWD problems of different heights (node rows: 16-row tiles, edge rows: 48-row tiles
    - one grid, gemm_mixed_kernel with two independent descriptors) == the two dosx_gemm launches: dz and the per-workgroup
    PReLU-slope partial rows bitwise (shapes outside the paired form - H 64, few edges - take the two launches inside)."""
    o = ops()
    from dostransformer_amd.functional import seg
    from dostransformer_amd.ops import EPI_PRELU_BWD
    outs = []
    for paired in (False, True):
        res, descs, alive = [], [], []                       # (a Seg holds a pointer, not the tensor)
        for i, M in enumerate((Mn, Me)):
            dy, z, W = rnd(M, H, seed=5 + i), rnd(M, H, seed=7 + i), rnd(H, H, seed=9 + i, scale=H ** -0.5)
            alpha = torch.tensor([0.25 + 0.1 * i], device=DEV)
            rows = o.gemm_partial_rows(M, H, EPI_PRELU_BWD)
            part, dz = torch.zeros(rows, 1, device=DEV), torch.empty(M, H, device=DEV)
            d = dict(M=M, N=H, segs=[seg(dy)], w=W, out=dz, w_layout=1, epi=EPI_PRELU_BWD, aux=z, epi_alpha=alpha, partials=part,
                     partial_ld=1)
            descs.append(d)
            alive += [dy, z, W, alpha]
            res += [dz, part]
            if i == 0:
                # reference for the first problem: da = dy W, dz = da * (z >= 0 ? 1 : alpha), dalpha = sum da * z over z < 0
                da = dy.double() @ W.double()
                ref_dz = torch.where(z >= 0, da, da * alpha.double())
                ref_al = float((da * z.double())[z < 0].sum())
        if paired:
            o.gemm_pair(descs[0], descs[1])
        else:
            o.gemm(**descs[0])
            o.gemm(**descs[1])
        torch.cuda.synchronize()
        outs.append(res)
    assert err(outs[0][0], ref_dz) < 1e-5 and abs(float(outs[0][1].sum()) - ref_al) < 1e-3 * max(1.0, abs(ref_al))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("S,B,H", [(51, 64, 128), (51, 3, 64), (201, 5, 256), (7, 1, 128), (201, 64, 256)])
def test_heads_backward_in_one_launch(S, B, H):
    """dosx_heads_bwd (round 6, csrc/heads.hip): the key-side LayerNorm backward + LeakyReLU backward of the heads' outputs and the
    two heads' input-gradient products (DOSTransformer_phonon.py:93-109 differentiated w.r.t. E1) as ONE launch, against the three
    launches it replaces (dosx_rownorm_bwd_act, two dosx_gemm with row-mapped A, the second accumulating) and float64."""
    o = ops()
    from dostransformer_amd.ops import rowmap, seg
    rows2 = S * 2 * B
    dkvs, kvs, ddosin, dosin = rnd(rows2, H, seed=1), rnd(rows2, H, seed=2), rnd(rows2, H, seed=3), rnd(rows2, H, seed=4)
    rstd = rnd(rows2, seed=5).abs() + 0.5
    Wfc, Wfp = rnd(H, 2 * H, seed=6, scale=H ** -0.5), rnd(H, 2 * H + H // 2, seed=7, scale=H ** -0.5)
    dpre0, dE0 = torch.full((rows2, H), float("nan"), device=DEV), torch.full((S * B, H), float("nan"), device=DEV)
    o.rownorm_bwd_act(dkvs, kvs, rstd, ddosin, dosin, 0.01, dpre0, rows2, H)
    map0, map1 = rowmap(d=B, m=2 * B, c=1, off=0), rowmap(d=B, m=2 * B, c=1, off=B)
    o.gemm(S * B, H, [seg(dpre0, rmap=map0)], Wfc[:, :H], dE0, w_layout=1)
    o.gemm(S * B, H, [seg(dpre0, rmap=map1)], Wfp[:, :H], dE0, w_layout=1, res=dE0)
    dpre1, dE1 = torch.full((rows2, H), float("nan"), device=DEV), torch.full((S * B, H), float("nan"), device=DEV)
    o.heads_bwd(S, B, H, dkvs, kvs, rstd, ddosin, dosin, 0.01, dpre1, Wfc, Wfp, dE1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dpre1).all()) and bool(torch.isfinite(dE1).all())
    assert err(dpre1, dpre0) < 2e-6 and err(dE1, dE0) < 1e-5
    # float64
    g, xh = dkvs.double(), kvs.double()
    t = ddosin.double() + rstd.double()[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    dp = torch.where(dosin.double() > 0, t, 0.01 * t).reshape(S, 2 * B, H)
    ref = (dp[:, :B] @ Wfc[:, :H].double() + dp[:, B:] @ Wfp[:, :H].double()).reshape(S * B, H)
    assert err(dpre1, dp.reshape(rows2, H)) < 2e-5 and err(dE1, ref) < 2e-5
