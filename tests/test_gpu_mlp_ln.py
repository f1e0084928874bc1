"""The one-launch NodeModel MLP (csrc/mlp2.hip: Linear -> LayerNorm -> PReLU -> Linear + residual, DOSTransformer_phonon.py:
200-212) against a plain torch fp32/fp64 reference of the same block and against the two-GEMM libdosx path it replaces."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a0, a1, w1, b1, g, b, al, w2, b2, res, dy):
    a0, a1 = a0.double().requires_grad_(True), a1.double().requires_grad_(True)
    ps = [t.double().requires_grad_(True) for t in (w1, b1, g, b, al, w2, b2)]
    w1, b1, g, b, al, w2, b2 = ps
    z = torch.cat([a0, a1], 1) @ w1.t() + b1
    y = torch.nn.functional.layer_norm(z, (z.shape[1],), g, b, 1e-5)
    y = torch.where(y >= 0, y, al * y)
    out = y @ w2.t() + b2 + (res.double() if res is not None else 0)
    out.backward(dy.double())
    return out.detach(), torch.cat([a0.grad, a1.grad], 1), [p.grad for p in ps]


@pytest.mark.parametrize("M,H", [(450, 128), (1, 128), (17, 64), (1554, 256), (33, 128)])
@pytest.mark.parametrize("with_res", [True, False])
def test_mlp_ln_fused_matches_reference(M, H, with_res, monkeypatch):
    from dostransformer_amd import functional as Fn
    from dostransformer_amd import ops
    from dostransformer_amd.ops import seg
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(M * 7 + H)
    r = lambda *s: (torch.randn(*s, generator=gen)).to(dev)
    x, agg = r(M, H), r(M, H)
    P = {"m.0.weight": r(2 * H, 2 * H) / (2 * H) ** 0.5, "m.0.bias": 0.1 * r(2 * H), "m.1.weight": 1 + 0.1 * r(2 * H),
         "m.1.bias": 0.1 * r(2 * H), "m.2.weight": torch.tensor([0.25], device=dev), "m.3.weight": r(H, 2 * H) / (2 * H) ** 0.5,
         "m.3.bias": 0.1 * r(H)}
    dy = r(M, H)
    res = x if with_res else None
    monkeypatch.setattr(ops, "MLP_LN_MAX_KN", 512 * 512)          # (the hidden-256 shape is off by default: slower there)
    assert ops.mlp_ln_supported(M, 2 * H, 2 * H, H)
    out_ref, dcat_ref, gp = _ref(x, agg, *[P[k] for k in ("m.0.weight", "m.0.bias", "m.1.weight", "m.1.bias", "m.2.weight",
                                                          "m.3.weight", "m.3.bias")], res, dy)

    def run(plain):
        G = {k: torch.zeros_like(v) for k, v in P.items()}
        a = Fn.SegList([seg(x), seg(agg)], [x, agg], plain=(x, agg) if plain else None)
        y, ctx = Fn.mlp_ln_fwd(P, "m", a, M, H, res=res)
        sink = ops.GradSink(torch.device(dev))
        dcat = Fn.mlp_ln_bwd(P, G, "m", ctx, dy, sink)
        sink.flush()
        sink.release()
        torch.cuda.synchronize()
        return y, dcat, G, ctx

    y1, d1, G1, c1 = run(True)
    y0, d0, G0, c0 = run(False)
    sc = lambda t: float(t.abs().max()) + 1e-30
    # against the fp64 reference
    assert float((y1.double() - out_ref).abs().max()) <= 2e-5 * sc(out_ref)
    assert float((d1.double() - dcat_ref).abs().max()) <= 2e-5 * sc(dcat_ref)
    for k, g in zip(("m.0.weight", "m.0.bias", "m.1.weight", "m.1.bias", "m.2.weight", "m.3.weight", "m.3.bias"), gp):
        assert float((G1[k].double() - g.reshape(G1[k].shape)).abs().max()) <= 5e-5 * sc(g), k
    # against the two-GEMM path (same arithmetic, different tiling: fp32 rounding only)
    assert float((y1 - y0).abs().max()) <= 5e-6 * sc(y0)
    assert float((d1 - d0).abs().max()) <= 5e-6 * sc(d0)
    assert float((c1[1] - c0[1]).abs().max()) <= 5e-6 * sc(c0[1])          # xhat
    assert float((c1[2] - c0[2]).abs().max()) <= 5e-6 * sc(c0[2])          # rstd
    for k in G1:
        assert float((G1[k] - G0[k]).abs().max()) <= 2e-5 * sc(G0[k]), k


def test_mlp_ln_fused_is_reproducible_and_row_local():
    """Bitwise run-to-run; rows of a tile do not see each other (a row computed alone equals the row in the batch)."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    H, M = 128, 100
    gen = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    x, agg = r(M, H), r(M, H)
    w1, b1, g, b, al, w2, b2 = r(2 * H, 2 * H) / 16, r(2 * H), 1 + 0.1 * r(2 * H), r(2 * H), torch.tensor([0.25], device=dev), r(H, 2 * H) / 16, r(H)

    def fwd(xx, aa):
        m = xx.shape[0]
        xh, rs, out = torch.empty(m, 2 * H, device=dev), torch.empty(m, device=dev), torch.empty(m, H, device=dev)
        ops.mlp_ln_fwd(m, xx, aa, w1, b1, g, b, al, w2, b2, xx, xh, rs, out)
        torch.cuda.synchronize()
        return out
    o1, o2 = fwd(x, agg), fwd(x, agg)
    assert torch.equal(o1, o2)
    o_row = fwd(x[37:38].contiguous(), agg[37:38].contiguous())
    assert torch.equal(o_row[0], o1[37])


@pytest.mark.parametrize("M,H", [(450, 128), (1554, 256), (17, 128), (1, 256)])
def test_mlp_ln_forward_also_multiplies_the_next_layers_node_products(M, H):
    """DosxMlpLn.w3 (round 5): pq = out . [Wa | Wb]^T of the NEXT message-passing layer's factored EdgeModel Linear (Wa, Wb: the
    first two H-column blocks of its [2H, 3H] weight) in the NodeModel launch == dosx_gemm_pair on the written output rows, to
    rounding; `out` itself, xhat, rstd bitwise the launch without the third product."""
    from dostransformer_amd import functional as Fn
    from dostransformer_amd import ops
    from dostransformer_amd.ops import seg
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(M * 3 + H)
    r = lambda *s: (torch.randn(*s, generator=gen)).to(dev)
    x, agg = r(M, H), r(M, H)
    P = {"m.0.weight": r(2 * H, 2 * H) / (2 * H) ** 0.5, "m.0.bias": 0.1 * r(2 * H), "m.1.weight": 1 + 0.1 * r(2 * H),
         "m.1.bias": 0.1 * r(2 * H), "m.2.weight": torch.tensor([0.25], device=dev), "m.3.weight": r(H, 2 * H) / (2 * H) ** 0.5,
         "m.3.bias": 0.1 * r(H)}
    W1n = r(2 * H, 3 * H) / (3 * H) ** 0.5
    assert ops.mlp_ln_fwd_supported(M, 2 * H, 2 * H, H)
    a = Fn.SegList([seg(x), seg(agg)], [x, agg], plain=(x, agg))
    y0, c0 = Fn.mlp_ln_fwd(P, "m", a, M, H, res=x)
    pq = torch.full((M, 4 * H), float("nan"), device=dev)
    y1, c1 = Fn.mlp_ln_fwd(P, "m", a, M, H, res=x, pq_next=(W1n, pq))
    ref = torch.empty(M, 4 * H, device=dev)
    ops.gemm_pair(dict(M=M, N=2 * H, segs=[seg(y0)], w=W1n[:, :H], out=ref[:, :2 * H]),
                  dict(M=M, N=2 * H, segs=[seg(y0)], w=W1n[:, H:2 * H], out=ref[:, 2 * H:]))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(c0[1], c1[1]) and torch.equal(c0[2], c1[2])
    assert not torch.isnan(pq).any()
    assert float((pq - ref).abs().max()) <= 5e-6 * float(ref.abs().max())
    exact = torch.cat([y0.double() @ W1n[:, :H].double().t(), y0.double() @ W1n[:, H:2 * H].double().t()], 1)
    assert float((pq.double() - exact).abs().max()) <= 2e-5 * float(exact.abs().max())


def _node_block(M, H, seed, dev="cuda:0"):
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=gen)).to(dev)
    x, agg = r(M, H), r(M, H)
    W = dict(w1=r(2 * H, 2 * H) / (2 * H) ** 0.5, b1=0.1 * r(2 * H), g=1 + 0.1 * r(2 * H), b=0.1 * r(2 * H),
             al=torch.tensor([0.25], device=dev), w2=r(H, 2 * H) / (2 * H) ** 0.5, b2=0.1 * r(H))
    return x, agg, W, r


@pytest.mark.parametrize("M,H", [(450, 128), (1, 128), (17, 64), (33, 128), (1000, 64), (255, 128), (16, 64)])
@pytest.mark.parametrize("with_res", [True, False])
def test_column_split_node_mlp_forward(M, H, with_res):
    """Round 6: the column-split NodeModel forward - hidden / 16 workgroups per 16-row tile, the pre-LayerNorm
    tile exchanged in-launch (publish / ticket / every sibling waits and reads back) - against the one-workgroup-per-tile kernel
    (fp32 rounding: the k range is split over 4 waves) and float64; with and without the third product (the next layer's node
    products, a second in-launch exchange of the output tile); launched repeatedly: same bits, counters back at zero."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    x, agg, W, r = _node_block(M, H, 11 * M + H)
    res = x if with_res else None
    W1n = r(2 * H, 3 * H) / (3 * H) ** 0.5
    assert ops.mlp_ln_cs(M, 2 * H, 2 * H, H)

    def fwd(cs, third):
        xh, rs, out = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev),
                       torch.full((M, H), float("nan"), device=dev))
        pq = torch.full((M, 4 * H), float("nan"), device=dev) if third else None
        ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], res, xh, rs, out,
                       w3=W1n if third else None, nb3=2 if third else 0, pq=pq, cs=cs)
        torch.cuda.synchronize()
        return out, xh, rs, pq
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref = fwd(False, False)
    for third in (False, True):
        got = fwd(True, third)
        for a_, b_ in zip(got[:3], ref[:3]):
            assert bool(torch.isfinite(a_).all())
            assert float((a_ - b_).abs().max()) <= 5e-6 * sc(b_)
        for _ in range(3):                          # counters are back at zero: the same launch again gives the same bits
            again = fwd(True, third)
            assert all(torch.equal(u, v) for u, v in zip(again[:3], got[:3]))
            assert not third or torch.equal(again[3], got[3])
        if third:
            exact = torch.cat([got[0].double() @ W1n[:, :H].double().t(), got[0].double() @ W1n[:, H:2 * H].double().t()], 1)
            assert float((got[3].double() - exact).abs().max()) <= 2e-5 * sc(exact)
    # float64
    z = torch.cat([x, agg], 1).double() @ W["w1"].double().t() + W["b1"].double()
    y = torch.nn.functional.layer_norm(z, (2 * H,), W["g"].double(), W["b"].double(), 1e-5)
    y = torch.where(y >= 0, y, 0.25 * y)
    o64 = y @ W["w2"].double().t() + W["b2"].double() + (res.double() if res is not None else 0)
    assert float((got[0].double() - o64).abs().max()) <= 2e-5 * sc(o64)


@pytest.mark.parametrize("M,H", [(450, 128), (1, 128), (17, 64), (33, 128), (1000, 64), (255, 128)])
@pytest.mark.parametrize("add_dy", [True, False])
def test_column_split_node_mlp_backward(M, H, add_dy):
    """... and the backward: dz, dcat (+ the residual connection's dy on its first H columns), the summed [dgamma | dbeta | dalpha]
    partial rows against the one-workgroup-per-tile kernel and against float64 autograd; dcat as a strided [M, 2H] view."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    x, agg, W, r = _node_block(M, H, 13 * M + H)
    dy = r(M, H)
    xh, rs, out = torch.empty(M, 2 * H, device=dev), torch.empty(M, device=dev), torch.empty(M, H, device=dev)
    ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows = ops.mlp_ln_bwd_partial_rows(M)
    pld = 4 * H + 4

    def bwd(cs):
        dz = torch.full((M, 2 * H), float("nan"), device=dev)
        dcat = torch.full((M, 2 * H), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        ops.mlp_ln_bwd(M, dy, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=add_dy, cs=cs)
        torch.cuda.synchronize()
        return dz, dcat, part[:, :4 * H].sum(0), part[:, pld - 1].sum()
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref, got = bwd(False), bwd(True)
    for a_, b_ in zip(got, ref):
        assert bool(torch.isfinite(a_).all())
        assert float((a_ - b_).abs().max()) <= 2e-5 * sc(b_)
    again = bwd(True)
    assert all(torch.equal(u, v) for u, v in zip(again, got))
    # float64 autograd
    xa = torch.cat([x, agg], 1).double().requires_grad_(True)
    g64, b64, al64 = (W[k].double().requires_grad_(True) for k in ("g", "b", "al"))
    z = xa @ W["w1"].double().t() + W["b1"].double()
    y = torch.nn.functional.layer_norm(z, (2 * H,), g64, b64, 1e-5)
    o = torch.where(y >= 0, y, al64 * y) @ W["w2"].double().t()
    o.backward(dy.double())
    dcat64 = xa.grad + (torch.cat([dy.double(), torch.zeros(M, H, dtype=torch.float64, device=dev)], 1) if add_dy else 0)
    assert float((got[1].double() - dcat64).abs().max()) <= 2e-5 * sc(dcat64)
    assert float((got[2][:2 * H].double() - g64.grad).abs().max()) <= 5e-5 * sc(g64.grad)
    assert float((got[2][2 * H:].double() - b64.grad).abs().max()) <= 5e-5 * sc(b64.grad)
    assert abs(float(got[3]) - float(al64.grad)) <= 5e-5 * max(1.0, abs(float(al64.grad)))


def test_column_split_exchange_under_a_bandwidth_hog():
    """The in-launch exchange (write-through stores, ticket, agent-scope poll, write-through read-back) 400 times back to back while
    a second stream streams 1 GiB copies through HBM (every XCD's L2 thrashed): every launch bitwise the first."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    M, H = 450, 128
    x, agg, W, r = _node_block(M, H, 77)
    W1n = r(2 * H, 3 * H) / (3 * H) ** 0.5
    dy = r(M, H)
    hog_a, hog_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
    side = torch.cuda.Stream()
    outs = []
    bad = torch.zeros(1, device=dev)
    first = None
    for it in range(400):
        if it % 8 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        xh, rs, out, pq = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev),
                           torch.full((M, H), float("nan"), device=dev), torch.full((M, 4 * H), float("nan"), device=dev))
        ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], x, xh, rs, out, w3=W1n, nb3=2, pq=pq, cs=True)
        dz, dcat, part = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M, 2 * H), float("nan"), device=dev),
                          torch.full((ops.mlp_ln_bwd_partial_rows(M), 4 * H + 4), float("nan"), device=dev))
        ops.mlp_ln_bwd(M, dy, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=True, cs=True)
        cur = (out, xh, pq, dz, dcat, part[:, :4 * H].clone(), part[:, -1].clone())
        if first is None:
            first = cur
        else:
            for u, v in zip(cur, first):
                bad += (u != v).any().float()          # compared on the device: no host sync in the loop
    torch.cuda.synchronize()
    assert float(bad) == 0.0
    assert all(bool(torch.isfinite(t).all()) for t in first)


@pytest.mark.parametrize("n,H,two,add_dy", [(450, 128, False, True), (450, 128, True, False), (37, 64, True, True), (16, 128, False, True),
                                            (3, 64, False, False), (1000, 64, False, True), (261, 128, True, True)])
def test_node_side_gradient_inside_the_column_split_backward_launch(n, H, two, add_dy):
    """DosxMlpLnBwd.pre = 1 (round 6): the node side of the later layer's factored input gradient (dosx_node_grad: source-node
    sums of dz, dx = res + res2 + aggS Wa + aggD Wb) as the front part of the column-split NodeModel backward launch - three
    in-launch exchanges - against the two launches: aggs, dx (= the block's dy), dz, dcat, the summed partial rows; isolated
    nodes, long source segments; twice (same bits, counters back at zero)."""
    import numpy as np
    from dostransformer_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(n + H)
    deg = rng.integers(0, 30, size=n)
    deg[rng.random(n) < 0.15] = 0
    deg[0] = 150 if n > 3 else 5                        # one long source segment (more than 64 edges: two id chunks per wave)
    E = int(deg.sum())
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)).to(dev)
    perm = torch.from_numpy(rng.permutation(E).astype(np.int32)).to(dev)
    x, agg, W, r = _node_block(n, H, 17 * n + H)
    W2 = 2 * H
    dzE, aggd, W0 = r(E, W2), r(n, W2), r(W2, 3 * H) / W2 ** 0.5
    dcat_prev = r(n, W2)
    res2 = r(n, H) if two else None
    xh, rs, out = torch.empty(n, W2, device=dev), torch.empty(n, device=dev), torch.empty(n, H, device=dev)
    ops.mlp_ln_fwd(n, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows, pld = ops.mlp_ln_bwd_partial_rows(n), 4 * H + 4

    def run(fused):
        aggs, dx = torch.full((n, W2), float("nan"), device=dev), torch.full((n, H), float("nan"), device=dev)
        dz, dcat = torch.full((n, W2), float("nan"), device=dev), torch.full((n, W2), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        pre = None
        if fused:
            pre = dict(kind="node_grad", dz=dzE, rowptr_src=rowptr, perm_src=perm, aggd=aggd, w=W0, res=dcat_prev[:, :H], res2=res2, aggs=aggs)
        else:
            ops.node_grad(n, H, dzE, rowptr, perm, aggd, W0, dcat_prev[:, :H], res2, aggs, dx)
        ops.mlp_ln_bwd(n, dx, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=add_dy, cs=True, pre=pre)
        torch.cuda.synchronize()
        return aggs, dx, dz, dcat, part[:, :4 * H].sum(0), part[:, pld - 1].sum()
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref, got = run(False), run(True)
    for name, a_, b_ in zip(("aggs", "dx", "dz", "dcat", "dgamma|dbeta", "dalpha"), got, ref):
        assert bool(torch.isfinite(a_).all()), name
        assert float((a_ - b_).abs().max()) <= 3e-5 * sc(b_), (name, float((a_ - b_).abs().max()) / sc(b_))
    again = run(True)
    assert all(torch.equal(u, v) for u, v in zip(again, got))
    exact = dcat_prev[:, :H].double() + got[0].double() @ W0[:, :H].double() + aggd.double() @ W0[:, H:W2].double()
    if two:
        exact = exact + res2.double()
    assert float((got[1].double() - exact).abs().max()) <= 2e-5 * sc(exact)


@pytest.mark.parametrize("n,H,B", [(450, 128, 64), (37, 64, 5), (16, 128, 1), (1000, 64, 100)])
def test_dense_key_backward_inside_the_column_split_backward_launch(n, H, B):
    """DosxMlpLnBwd.pre = 2 (round 6): dosx_dense_normalize_pool_bwd - the to_dense_batch / key-LayerNorm backward plus the
    pooled decoder gradient, ghost nodes included - as the front part of the last layer's NodeModel backward launch: dx, dz,
    dcat, partial rows bitwise the two launches (row-local: no exchange, the same arithmetic)."""
    import numpy as np
    from dostransformer_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(n + H + B)
    x, agg, W, r = _node_block(n, H, 19 * n + H)
    n_ghost = min(5, n // 4)
    n_real = n - n_ghost
    sizes = np.diff(np.sort(np.concatenate([[0, n_real], rng.integers(0, n_real + 1, size=B - 1)])))
    node_graph = np.concatenate([np.repeat(np.arange(B), sizes), np.full(n_ghost, B)]).astype(np.int32)
    nmax = int(sizes.max())
    pos = np.concatenate([np.arange(s) for s in sizes] + [np.zeros(n_ghost, dtype=np.int64)])
    dense_row = np.where(node_graph < B, pos * B + np.minimum(node_graph, B - 1), nmax * B).astype(np.int32)
    dense_row_t, node_graph_t = torch.from_numpy(dense_row).to(dev), torch.from_numpy(node_graph).to(dev)
    dkv, kvhat = r(nmax * B + 1, H), r(nmax * B + 1, H)
    rstd_n = r(n).abs() + 0.5
    Kd = 2 * H
    dpool = r(B, Kd)
    xh, rs, out = torch.empty(n, 2 * H, device=dev), torch.empty(n, device=dev), torch.empty(n, H, device=dev)
    ops.mlp_ln_fwd(n, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows, pld = ops.mlp_ln_bwd_partial_rows(n), 4 * H + 4

    def run(fused):
        dx = torch.full((n, H), float("nan"), device=dev)
        dz, dcat = torch.full((n, 2 * H), float("nan"), device=dev), torch.full((n, 2 * H), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        pre = None
        if fused:
            pre = dict(kind="dense", dkv=dkv, kvhat=kvhat, rstd_nodes=rstd_n, dense_row=dense_row_t, dpool_ptr=dpool.data_ptr() + 4 * (Kd - H),
                       ld_dpool=Kd, node_graph=node_graph_t, num_graphs=B, ghost_row=nmax * B)
        else:
            ops.dense_normalize_pool_bwd(dkv, kvhat, rstd_n, dense_row_t, dpool.data_ptr() + 4 * (Kd - H), Kd, node_graph_t, B, dx, n, H,
                                         False, ghost_row=nmax * B)
        ops.mlp_ln_bwd(n, dx, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=True, cs=True, pre=pre)
        torch.cuda.synchronize()
        return dx, dz, dcat, part[:, :4 * H].clone(), part[:, pld - 1].clone()
    ref, got = run(False), run(True)
    sc = lambda t: float(t.abs().max()) + 1e-30
    for name, a_, b_ in zip(("dx", "dz", "dcat", "partials", "dalpha"), got, ref):
        assert bool(torch.isfinite(a_).all()), name
        assert float((a_ - b_).abs().max()) <= 2e-6 * sc(b_), name          # (row sums over 16 instead of 64 lanes: rounding)
    assert float(got[0][n_real:].abs().max()) == 0.0 if n_ghost else True    # ghost nodes: zero gradient
