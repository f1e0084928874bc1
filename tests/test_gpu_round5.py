"""Round 5: the ADVICE r4 fixes and the new paths of the round (see the individual tests)."""
import ctypes as C

import numpy as np
import pytest
import torch

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


def test_promotion_into_a_bucket_that_a_batch_object_filled():
    """a bucket first filled by ``step(batch)`` has no collate scratch; when ``step_dataset`` later PROMOTES
    a smaller shape into it, the scratch must be sized from the host slot (the collate kernels write node_row / edge_row up to
    the slot's padded counts), not from the requested bucket.  Same losses as a trainer without promotion."""
    import copy
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer
    cs = synth.phonon_crystals(40, seed=93, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    nmax = max(int(c["x"].shape[0]) for c in cs)
    order = sorted(range(40), key=lambda i: int(cs[i]["x"].shape[0]))
    big, small = order[-8:], order[:8]
    torch.manual_seed(3)
    m_a = DOSTransformer_phonon(3, 1, 118, 4, 32, DEV, 0.0).to(DEV)
    m_b = DOSTransformer_phonon(3, 1, 118, 4, 32, DEV, 0.0)
    m_b.load_state_dict(copy.deepcopy(m_a.state_dict()))
    m_b = m_b.to(DEV)
    ta = Trainer(m_a, lr=1e-3, replay=True, bucket=(8, 64), promote=10.0)
    tb = Trainer(m_b, lr=1e-3, replay=True, bucket=(8, 64))
    # the big bucket comes into being through a batch OBJECT (no collate scratch on the slot) ...
    gb = ds.collate(big, n_max=nmax)
    la, lb = ta.step(gb), tb.step(gb)
    assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb)))
    slot = next(iter(ta._slots.values()))
    assert getattr(slot, "scratch", None) is None
    # ... and the small shape is promoted into it on its first sighting
    la, lb = ta.step_dataset(ds, small, n_max=nmax), tb.step_dataset(ds, small, n_max=nmax)
    assert ta.slot_promoted == 1
    assert slot.scratch["node_row"].numel() == slot.g.meta.num_nodes and slot.scratch["edge_row"].numel() == slot.g.meta.num_edges
    assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb)))
    la, lb = ta.step_dataset(ds, small, n_max=nmax), tb.step_dataset(ds, small, n_max=nmax)
    assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb)))
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        if a.is_floating_point():
            assert torch.isfinite(a).all(), k
            assert float((a - b).abs().max()) < 5e-3, k


def _graph(n, seed, fat=()):
    """A random destination-sorted graph on n nodes: (src, dst, rowptr, deg, seg_tile) on the device; `fat`: in-degrees forced on
    the first nodes (over-full nodes -> chunk tiles)."""
    from dostransformer_amd.batch import seg_tiles_host
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 30, size=n)
    deg[rng.random(n) < 0.15] = 0
    for i, d in enumerate(fat):
        deg[i] = d
    E = int(deg.sum())
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    dst = torch.from_numpy(np.repeat(np.arange(n), deg).astype(np.int32)).to(DEV)
    src = torch.from_numpy(rng.integers(0, n, size=E).astype(np.int32)).to(DEV)
    tiles = torch.from_numpy(seg_tiles_host(rowptr)).to(DEV)
    return src, dst, torch.from_numpy(rowptr.astype(np.int32)).to(DEV), deg, tiles, E


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


@pytest.mark.parametrize("n,H,fat,mean", [(300, 128, (), False), (60, 128, (60, 96, 200), True), (45, 64, (49,), False), (8, 32, (), True)])
def test_prelu_layernorm_backward_on_node_aligned_tiles_also_sums_dz_per_node(n, H, fat, mean):
    """DOSX_EPI_PRELU_LN_BWD_SEG (round 5): the EdgeModel's second-Linear input gradient with the PReLU / LayerNorm backward in
    its epilogue, on the node-aligned row tiles of the message GEMM, ALSO leaves the destination-node sums of dz (what the
    factored first Linear's weight / input gradients are made of): dz and the summed partial rows against the row-block form
    (EPI_PRELU_LN_BWD), the node sums against dosx_segment_reduce on that dz; over-full nodes (chunk tiles + ticket) and
    isolated nodes included; twice (counters back at zero, bitwise repeatable)."""
    o = ops()
    src, dst, rp, deg, tiles, E = _graph(n, 7 * n + H, fat)
    W2 = 2 * H
    dy, W3 = rnd(E, H, seed=1), rnd(H, W2, seed=2, scale=H ** -0.5)
    xhat, rstd = rnd(E, W2, seed=3), rnd(E, seed=4).abs() + 0.5
    gam, bet, alpha = rnd(W2, seed=5), rnd(W2, seed=6), torch.tensor([0.25], device=DEV)
    scale = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(DEV) if mean else None
    pld = 2 * W2 + 4
    rows0 = o.gemm_partial_rows(E, W2, o.EPI_PRELU_LN_BWD)
    dz0, part0 = torch.empty(E, W2, device=DEV), torch.empty(rows0, pld, device=DEV)
    o.gemm(E, W2, [o.seg(dy)], W3, dz0, w_layout=1, epi=o.EPI_PRELU_LN_BWD, aux=xhat, aux_stats=rstd, epi_gamma=gam, epi_beta=bet,
           epi_alpha=alpha, partials=part0, partial_ld=pld)
    agg0 = torch.empty(n, W2, device=DEV)
    o.segment_reduce(dz0, rp, scale, agg0, None, None, n, E, W2)
    T = tiles.shape[1] - 1
    prev = None
    for rep in range(2):
        dz1 = torch.full((E, W2), float("nan"), device=DEV)
        part1 = torch.full((T, pld), float("nan"), device=DEV)
        agg1 = torch.full((n, W2), float("nan"), device=DEV)
        o.gemm(E, W2, [o.seg(dy)], W3, dz1, w_layout=1, epi=o.EPI_PRELU_LN_BWD_SEG, aux=xhat, aux_stats=rstd, epi_gamma=gam,
               epi_beta=bet, epi_alpha=alpha, partials=part1, partial_ld=pld, seg_tile=tiles, seg_rowptr=rp, seg_scale=scale,
               seg_agg=agg1)
        torch.cuda.synchronize()
        assert torch.equal(dz1, dz0)                             # the same products and row epilogue, whatever the tiling
        assert bool(torch.isfinite(agg1).all())
        assert float((agg1 - agg0).abs().max()) <= 4e-6 * float(agg0.abs().max() + 1e-6)
        p0, p1 = part0.double().sum(0), part1.double().sum(0)
        assert err(p1[:2 * W2], p0[:2 * W2]) < 2e-5 and abs(float(p1[-1] - p0[-1])) < 2e-5 * (abs(float(p0[-1])) + 1.0)
        if prev is not None:
            assert torch.equal(agg1, prev[0]) and torch.equal(part1[:, :2 * W2], prev[1][:, :2 * W2])
        prev = (agg1, part1)


@pytest.mark.parametrize("M,H", [(450, 128), (1554, 256), (33, 32), (9000, 64)])
def test_gemm_with_one_weight_block_per_k_segment(M, H):
    """DosxGemm.w_seg_off (round 5): out = [S | D] . [Wa ; Wb] + res with Wa = W[:, :H], Wb = W[:, H:2H] two column blocks of ONE
    [2H, 3H] matrix - the node part of the factored EdgeModel input gradient, dx = S Wa + D Wb, as one launch."""
    o = ops()
    S, D, W = rnd(M, 2 * H, seed=1), rnd(M, 2 * H, seed=2), rnd(2 * H, 3 * H, seed=3, scale=(2 * H) ** -0.5)
    res = rnd(M, 2 * H, seed=4)
    out = torch.full((M, H), float("nan"), device=DEV)
    o.gemm(M, H, [o.seg(S), o.seg(D)], W[:, :H], out, w_layout=1, w_seg_off=H, res=res[:, :H])
    torch.cuda.synchronize()
    ref = S.double() @ W[:, :H].double() + D.double() @ W[:, H:2 * H].double() + res[:, :H].double()
    assert err(out, ref) < 2e-5
    with pytest.raises(Exception):                                 # segment widths must be multiples of 32
        o.gemm(M, H, [o.seg(S, width=2 * H - 4), o.seg(D)], W[:, :H], out, w_layout=1, w_seg_off=H)


@pytest.mark.parametrize("M", [450, 17])
def test_node_mlp_backward_adds_the_residual_path(M):
    """DosxMlpLnBwd.add_dy (round 5): dcat[:, :H] += dy - the NodeModel's residual connection x' = x + MLP(cat[x, agg])
    (DOSTransformer_phonon.py:83,204-212) differentiated inside the one-launch backward; everything else bit for bit."""
    o = ops()
    H = 128
    dy, xhat, rstd = rnd(M, H, seed=1), rnd(M, 2 * H, seed=2), rnd(M, seed=3).abs() + 0.5
    w1, w2 = rnd(2 * H, 2 * H, seed=4, scale=0.06), rnd(H, 2 * H, seed=5, scale=0.06)
    gam, bet, alpha = rnd(2 * H, seed=6), rnd(2 * H, seed=7), torch.tensor([0.25], device=DEV)
    rows = o.mlp_ln_bwd_partial_rows(M)
    res = []
    for add in (False, True):
        dz, dcat = torch.empty(M, 2 * H, device=DEV), torch.empty(M, 2 * H, device=DEV)
        part = torch.empty(rows, 4 * H + 4, device=DEV)
        o.mlp_ln_bwd(M, dy, xhat, rstd, w1, w2, gam, bet, alpha, dz, dcat, part, add_dy=add)
        res.append((dz, dcat, part))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2][:, :4 * H], res[1][2][:, :4 * H])
    assert torch.equal(res[0][1][:, H:], res[1][1][:, H:])
    assert torch.equal(res[1][1][:, :H], res[0][1][:, :H] + dy)


@pytest.mark.parametrize("kind,H", [("phonon", 128), ("phonon", 64), ("edos", 64), ("edos", 256)])
def test_fused_factored_edge_layer_equals_the_plain_one(kind, H):
    """The factored EdgeModel first Linear with its gathers and node sums INSIDE the GEMM epilogues (round 5: add_p / add_q in
    EPI_LN, EPI_PRELU_LN_BWD_SEG, one w_seg_off GEMM for the node part of the input gradient, the NodeModel's residual inside its
    one-launch backward) against the gathered-concat form AND against round 4's factored form with stand-alone row kernels:
    outputs and every gradient of a training step agree to rounding; eager and replay give the same bits; ghost-padded batch
    with over-full nodes."""
    from dostransformer_amd import functional as Fn, synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    from dostransformer_amd.train import Trainer
    from tests.test_gpu_round3 import _fat_crystals
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        mk = lambda: DOSTransformer_phonon(3, 1, 118, 4, H, DEV, 0.0)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 1, 200, 41, 2, H, DEV, 0.0)
    g = collate(_fat_crystals(kind, 6, 5, torch.float32))
    gp = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 16, 256)).to(DEV)
    m0 = mk()
    sd0 = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    grads, params, outs = {}, {}, {}
    saved = (Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED, Fn._EDGE_ONE_LAUNCH, Fn._EDGE_ONE_LAUNCH_BWD)
    try:
        # plain: gathered-concat GEMM; rowkernels: round 4's factored form; epilogues: gathers / node sums inside dosx_gemm;
        # fused: the shipped default - hidden <= 128: EdgeModel forward and backward one launch each (csrc/edge_mlp.hip)
        for form in ("plain", "rowkernels", "epilogues", "fused"):
            Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED = form != "plain", 0.0, form in ("epilogues", "fused")
            Fn._EDGE_ONE_LAUNCH = Fn._EDGE_ONE_LAUNCH_BWD = form == "fused"
            for replay in (False, True):
                model = mk()
                model.load_state_dict(sd0)
                model = model.to(DEV)
                tr = Trainer(model, lr=1e-3, replay=replay)
                tr.forward_backward(gp)
                torch.cuda.synchronize()
                grads[(form, replay)] = {k: v.clone() for k, v in model.flat_params().G.items()}
                outs[(form, replay)] = [t.clone() for t in tr.last_outputs]
                for _ in range(2):
                    tr.step(gp)
                torch.cuda.synchronize()
                params[(form, replay)] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    finally:
        Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED, Fn._EDGE_ONE_LAUNCH, Fn._EDGE_ONE_LAUNCH_BWD = saved
    n_real, n_pad = g.meta.num_nodes, gp.meta.num_nodes
    for other in ("plain", "rowkernels", "epilogues"):
        for u, v in zip(outs[("fused", False)], outs[(other, False)]):
            if u.shape[0] == n_pad:
                u, v = u[:n_real], v[:n_real]
            assert err(u, v) < 5e-6, other
        for k, v in grads[(other, False)].items():
            # H >= 128: an activation gate whose pre-activation is below fp32 resolution may flip between two summation orders and
            # moves one row of a few tensors by 1e-4 .. 1e-3 of the maximum (tools/grad_errors.py): the bound on the
            # maximum is loose there, the 99th percentile of the element errors is what guards; a one-element gradient (a PReLU
            # slope) is ONE long sum with cancellation
            u = grads[("fused", False)][k]
            if H >= 128 or v.numel() == 1:
                assert err(u, v) < 3e-3, (other, k)
                if v.numel() >= 1000:
                    q = torch.quantile(((u.double() - v.double()).abs() / (v.double().abs().max() + 1e-12)).flatten()[:4_000_000], 0.99)
                    assert float(q) < 1e-4, (other, k, float(q))
            else:
                assert err(u, v) < 1e-4, (other, k)
    for k in params[("fused", False)]:
        assert torch.equal(params[("fused", False)][k], params[("fused", True)][k]), ("eager vs replay", k)


@pytest.mark.parametrize("n,H,fat,mean,last", [(300, 128, (), True, False), (60, 128, (60, 96, 200), False, False),
                                               (45, 64, (49, 48), True, True), (5, 64, (), False, False), (700, 128, (97,), True, True)])
def test_edge_model_forward_in_one_launch(n, H, fat, mean, last):
    """dosx_edge_mlp_fwd (round 5, csrc/edge_mlp.hip): the EdgeModel with its first Linear factored + scatter_mean / scatter_sum
    + the edge residual (DOSTransformer_phonon.py:186-197,209,84) as ONE launch on the node-aligned row tiles - against the two
    dosx_gemm launches it replaces (EPI_LN with gathered addends, then PRO_LN_PRELU + EPI_SEGSUM) and against float64; over-full
    and isolated nodes; with / without the edge update (the last layer's is dead); twice (counters back at zero, same bits)."""
    from dostransformer_amd import functional as Fn
    o = ops()
    src, dst, rp, deg, tiles, E = _graph(n, 11 * n + H, fat)
    gen = torch.Generator().manual_seed(n)
    P = {"k.0.weight": torch.randn(2 * H, 3 * H, generator=gen) * (3 * H) ** -0.5, "k.0.bias": torch.randn(2 * H, generator=gen),
         "k.1.weight": torch.randn(2 * H, generator=gen), "k.1.bias": torch.randn(2 * H, generator=gen),
         "k.2.weight": torch.tensor([0.25]), "k.3.weight": torch.randn(H, 2 * H, generator=gen) * (2 * H) ** -0.5,
         "k.3.bias": torch.randn(H, generator=gen)}
    P = Fn.pack_params({k: v.to(DEV) for k, v in P.items()})        # (one buffer window for the two weight matrices)
    x, e = torch.randn(n, H, generator=gen).to(DEV), torch.randn(E, H, generator=gen).to(DEV)
    scale = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(DEV) if mean else None

    class M_:           # what functional.mlp_ln_fwd reads of a GraphMeta
        pass
    m = M_()
    m.src, m.dst, m.num_nodes, m.num_edges, m.seg_tile, m.rowptr_dst = src, dst, n, E, tiles, rp
    res = {}
    saved = (Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED, Fn._EDGE_ONE_LAUNCH)
    try:
        for one in (False, True, True):
            Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED, Fn._EDGE_ONE_LAUNCH = 0.0, True, one
            a = Fn.SegList([o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(e)], [x, e])
            a.factor = (x, e, m)
            agg = torch.full((n, H), float("nan"), device=DEV)
            e_out = None if last else torch.full((E, H), float("nan"), device=DEV)
            _, ctx = Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(tiles, rp, scale, agg, e, e_out))
            torch.cuda.synchronize()
            if one and True in res:
                r = res[True]
                assert torch.equal(agg, r[0]) and torch.equal(ctx[1], r[2]) and (last or torch.equal(e_out, r[1]))
            res[one] = (agg, e_out, ctx[1].clone(), ctx[2].clone())
    finally:
        Fn._FACTOR_MIN_GF, Fn._FACTOR_FUSED, Fn._EDGE_ONE_LAUNCH = saved
    (agg0, e0, xh0, rs0), (agg1, e1, xh1, rs1) = res[False], res[True]
    assert bool(torch.isfinite(agg1).all()) and bool(torch.isfinite(xh1).all())
    assert err(xh1, xh0) < 1e-5 and err(rs1, rs0) < 1e-5
    assert float((agg1 - agg0).abs().max()) <= 1e-5 * float(agg0.abs().max() + 1e-6)
    if not last:
        assert err(e1, e0) < 1e-5
    # float64
    W1, W3 = P["k.0.weight"].double(), P["k.3.weight"].double()
    z = torch.cat([x[src.long()], x[dst.long()], e], 1).double() @ W1.T + P["k.0.bias"].double()
    mu, var = z.mean(1, keepdim=True), z.var(1, unbiased=False, keepdim=True)
    xh = (z - mu) * (var + 1e-5).rsqrt()
    y = xh * P["k.1.weight"].double() + P["k.1.bias"].double()
    msg = torch.where(y < 0, 0.25 * y, y) @ W3.T + P["k.3.bias"].double()
    ref = torch.zeros(n, H, dtype=torch.float64, device=DEV).index_add_(0, dst.long(), msg)
    if mean:
        ref = ref * scale.double()[:, None]
    assert err(xh1, xh) < 2e-5 and err(agg1, ref) < 2e-5
    if not last:
        assert err(e1, e.double() + msg) < 2e-5


@pytest.mark.parametrize("n,H,fat,mean,last", [(300, 128, (), True, False), (60, 128, (60, 96, 200), False, False),
                                               (45, 64, (49, 48), True, True), (5, 64, (), False, True), (700, 128, (97,), True, False)])
def test_edge_model_backward_in_one_launch(n, H, fat, mean, last):
    """dosx_edge_mlp_bwd (round 5, csrc/edge_mlp.hip) against the three launches it replaces - dosx_edge_grad_combine, dosx_gemm
    with EPI_PRELU_LN_BWD_SEG, the E-row input-gradient GEMM dz Wc + de_next: message gradient, dz, the destination-node sums,
    de, the summed parameter-gradient partial rows; over-full / isolated nodes; last layer (no incoming edge-state gradient);
    twice (counters back at zero, same bits)."""
    o = ops()
    src, dst, rp, deg, tiles, E = _graph(n, 13 * n + H, fat)
    W2 = 2 * H
    dcat_n, de_next = rnd(n, W2, seed=1), (None if last else rnd(E, H, seed=2))
    dagg = dcat_n[:, H:]
    xhat, rstd = rnd(E, W2, seed=3), rnd(E, seed=4).abs() + 0.5
    gam, bet, alpha = rnd(W2, seed=5), rnd(W2, seed=6), torch.tensor([0.25], device=DEV)
    # (the kernel addresses both weight matrices through one 2 GiB buffer window: one allocation, as in the models' flat buffer)
    wflat = torch.empty(H * W2 + W2 * 3 * H, device=DEV)
    W3, W1 = wflat[:H * W2].view(H, W2), wflat[H * W2:].view(W2, 3 * H)
    W3.copy_(rnd(H, W2, seed=7, scale=H ** -0.5))
    W1.copy_(rnd(W2, 3 * H, seed=8, scale=W2 ** -0.5))
    scale = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(DEV) if mean else None
    pld = 2 * W2 + 4
    T = tiles.shape[1] - 1
    # the three-launch form
    dmsg0 = torch.empty(E, H, device=DEV)
    o.edge_grad_combine(de_next, dcat_n.data_ptr() + 4 * H, W2, dst, scale, dmsg0, E, H)
    dz0, part0, agg0 = torch.empty(E, W2, device=DEV), torch.empty(T, pld, device=DEV), torch.empty(n, W2, device=DEV)
    o.gemm(E, W2, [o.seg(dmsg0)], W3, dz0, w_layout=1, epi=o.EPI_PRELU_LN_BWD_SEG, aux=xhat, aux_stats=rstd, epi_gamma=gam, epi_beta=bet,
           epi_alpha=alpha, partials=part0, partial_ld=pld, seg_tile=tiles, seg_rowptr=rp, seg_agg=agg0)
    de0 = torch.empty(E, H, device=DEV)
    o.gemm(E, H, [o.seg(dz0)], W1[:, W2:], de0, w_layout=1, res=de_next)
    prev = None
    for rep in range(2):
        dmsg1, dz1, de1 = (torch.full((E, w), float("nan"), device=DEV) for w in (H, W2, H))
        part1, agg1 = torch.full((T, pld), float("nan"), device=DEV), torch.full((n, W2), float("nan"), device=DEV)
        o.edge_mlp_bwd(E, H, dagg, de_next, dst, xhat, rstd, W3, W1[:, W2:], gam, bet, alpha, dmsg1, dz1, de1, part1, tiles, rp, scale, agg1)
        torch.cuda.synchronize()
        assert err(dmsg1, dmsg0) < 1e-6
        assert err(dz1, dz0) < 2e-5 and err(de1, de0) < 2e-5
        assert bool(torch.isfinite(agg1).all())
        assert float((agg1 - agg0).abs().max()) <= 2e-5 * float(agg0.abs().max() + 1e-6)
        p0, p1 = part0.double().sum(0), part1.double().sum(0)
        assert err(p1[:2 * W2], p0[:2 * W2]) < 2e-5 and abs(float(p1[-1] - p0[-1])) < 2e-5 * (abs(float(p0[-1])) + 1.0)
        if prev is not None:
            assert all(torch.equal(u, v) for u, v in zip(prev, (dmsg1, dz1, de1, agg1)))
        prev = (dmsg1, dz1, de1, agg1)


@pytest.mark.parametrize("n,H,two", [(450, 128, False), (1554, 256, True), (37, 64, True), (16, 128, False), (3, 64, False)])
def test_node_side_of_the_factored_input_gradient_in_one_launch(n, H, two):
    """dosx_node_grad (round 5): aggS = source-node sums of dz (CSR by source, isolated nodes included) and
    dx = res (+ res2) + aggS Wa + aggD Wb in one launch, against dosx_segment_reduce_perm + float64 products."""
    o = ops()
    rng = np.random.default_rng(n + H)
    deg = rng.integers(0, 30, size=n)
    deg[rng.random(n) < 0.15] = 0
    E = max(int(deg.sum()), 1)
    if deg.sum() == 0:
        deg[0] = 1
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)).to(DEV)
    perm = torch.from_numpy(rng.permutation(E).astype(np.int32)).to(DEV)
    W2 = 2 * H
    dz, aggd, W = rnd(E, W2, seed=1), rnd(n, W2, seed=2), rnd(W2, 3 * H, seed=3, scale=W2 ** -0.5)
    dcat = rnd(n, W2, seed=4)
    res2 = rnd(n, H, seed=5) if two else None
    aggs0 = torch.empty(n, W2, device=DEV)
    o.segment_reduce_perm(dz, rowptr, perm, aggs0, n, E, W2)
    aggs1, dx1 = torch.full((n, W2), float("nan"), device=DEV), torch.full((n, H), float("nan"), device=DEV)
    o.node_grad(n, H, dz, rowptr, perm, aggd, W, dcat[:, :H], res2, aggs1, dx1)
    torch.cuda.synchronize()
    assert float((aggs1 - aggs0).abs().max()) <= 2e-6 * float(aggs0.abs().max() + 1e-6)
    ref = dcat[:, :H].double() + aggs0.double() @ W[:, :H].double() + aggd.double() @ W[:, H:W2].double()
    if two:
        ref = ref + res2.double()
    assert err(dx1, ref) < 2e-5


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


@pytest.mark.parametrize("Sq,Bq,Nk,Bk,H,bcast", [(201, 6, 41, 3, 256, False), (201, 5, 64, 5, 256, False), (51, 4, 12, 4, 128, False),
                                                   (51, 6, 9, 3, 64, False), (70, 3, 48, 3, 128, False), (33, 300, 17, 150, 64, False),
                                                   (201, 140, 16, 70, 128, False), (7, 3, 1, 3, 256, False), (51, 5, 7, 5, 128, True),
                                                   (64, 9, 33, 9, 256, True)])
@pytest.mark.parametrize("drop", [0.0, 0.35])
def test_attention_on_crystal_aligned_tiles(Sq, Bq, Nk, Bk, H, bcast, drop):
    """csrc/attention_aligned.hip behind dosx_attention_fwd / dosx_attention_bwd (one-launch form): <= 64 keys, hidden 64 / 128 /
    256, workgroups that own one, several (Bq = 140: 2 per crystal) or all (Bq = 300) query tiles of a crystal, broadcast query
    rows, dropout masks - against the float64 reference of multihead_attention.py:49-76 + the LayerNorm / residual around it;
    repeated launches bitwise equal (counters back at zero); with the mode switched off the same call takes attention.hip's
    kernels and agrees to rounding."""
    from tests.test_gpu_ops import _attn_ref
    from dostransformer_amd import _lib
    from dostransformer_amd._lib import Attn
    o = ops()
    lib = _lib.load()
    qs, qb = (1, 0) if bcast else (Bq, 1)
    x = rnd(Sq if bcast else Sq * Bq, H, seed=1).double().requires_grad_(True)
    kv = rnd(Nk * Bk, H, seed=2)
    kv[-Bk:] = 0                        # zero-padded atoms
    kv = kv.double().requires_grad_(True)
    gam, bet = rnd(H, seed=3).double().requires_grad_(True), (0.3 * rnd(H, seed=4)).double().requires_grad_(True)
    mask = None
    if drop > 0:
        mask = (torch.rand(Bq, Sq, Nk, generator=torch.Generator().manual_seed(9)) >= drop).float().to(DEV) / (1 - drop)
    ref, pref = _attn_ref(x, kv, gam, bet, Sq, Bq, Nk, Bk, H, qs, qb, None if mask is None else mask.double())
    dout = rnd(Sq * Bq, H, seed=5)
    ref.backward(dout.double())
    f = lambda t: t.detach().float().contiguous()
    xf, kvf, gf, bf = f(x), f(kv), f(gam), f(bet)
    nqt, nkt = (Sq + 31) // 32, (Nk + 15) // 16
    base = rnd(Nk * Bk, H, seed=6)
    g1, b1 = 1 + 0.2 * rnd(H, seed=7), 0.3 * rnd(H, seed=8)

    def run(mode):
        prev = lib.dosx_attention_aligned_mode(mode)
        try:
            a = Attn()
            a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bk, H, qs, qb
            out, probs = torch.full((Sq * Bq, H), float("nan"), device=DEV), torch.full((Bq, Sq, Nk), float("nan"), device=DEV)
            qstats, ostats = torch.full((Sq * Bq, 2), float("nan"), device=DEV), torch.full((Sq * Bq, 2), float("nan"), device=DEV)
            a.x, a.kvhat, a.gamma0, a.beta0 = xf.data_ptr(), kvf.data_ptr(), gf.data_ptr(), bf.data_ptr()
            a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qstats.data_ptr(), ostats.data_ptr()
            a.drop_mask = mask.data_ptr() if mask is not None else None
            ln1 = torch.full((Sq * Bq, H), float("nan"), device=DEV)      # DosxAttn.ln1_*: the layer's next LayerNorm on the output rows
            a.ln1_gamma, a.ln1_beta, a.ln1_out = g1.data_ptr(), b1.data_ptr(), ln1.data_ptr()
            o.attention_fwd(a)
            assert err(ln1, torch.nn.functional.layer_norm(ref.detach(), (H,), g1.double(), b1.double(), 1e-5)) < 5e-5
            dx = torch.full((Sq * Bq, H), float("nan"), device=DEV)
            dkv = base.clone()
            part = torch.full((Bq * nqt + Bk * nkt, 2 * H), float("nan"), device=DEV)
            kvp = torch.full((Bq * nqt * Nk, H), float("nan"), device=DEV)
            a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), None, dkv.data_ptr(), 1
            a.partials_q, a.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * H
            a.dkv_part, a.dkv_cnt = kvp.data_ptr(), o.COUNTERS.take(DEV, Bk)
            outs = []
            for _ in range(2):
                dkv.copy_(base)
                o.attention_bwd(a)
                torch.cuda.synchronize()
                outs.append((dx.clone(), dkv.clone(), part.clone()))
            assert all(torch.equal(u, v) for u, v in zip(*outs))
            return out, probs, qstats, ostats, dx, dkv - base, part
        finally:
            lib.dosx_attention_aligned_mode(prev)

    new, old = run(2), run(0)
    for out, probs, qstats, ostats, dx, dkv, part in (new, old):
        assert err(out, ref) < 3e-5 and err(probs, pref) < 3e-5
        assert err(ostats[:, 0], ref.detach().mean(1)) < 5e-5
        assert err(ostats[:, 1], 1 / torch.sqrt(ref.detach().var(1, unbiased=False) + 1e-5)) < 5e-5
        ps = part.double().sum(0)
        dxr = dx.double().reshape(Sq, Bq, H).sum(1) if bcast else dx
        assert err(dxr, x.grad) < 5e-5 and err(dkv, kv.grad) < 5e-5
        assert err(ps[:H], gam.grad) < 5e-5 and err(ps[H:], bet.grad) < 5e-5
    assert err(new[2], old[2]) < 1e-5               # the LayerNorm-0 statistics of the query rows


@pytest.mark.parametrize("Sq,Bq,Nk,H", [(201, 3, 201, 256), (70, 3, 70, 128), (51, 4, 12, 128)])
def test_attention_forward_also_writes_the_next_layernorm(Sq, Bq, Nk, H):
    """DosxAttn.ln1_out on attention.hip's kernels (more than 64 keys: the 201-key Electron-DOS self attention; mode 0 for the
    small shape): LN1 of the output rows == F.layer_norm of the rows the same call writes; refused beyond 320 keys."""
    from dostransformer_amd import _lib
    from dostransformer_amd._lib import Attn, DosxError
    o = ops()
    lib = _lib.load()
    x, kv, gam, bet = rnd(Sq * Bq, H, seed=1), rnd(Nk * Bq, H, seed=2), rnd(H, seed=3), 0.3 * rnd(H, seed=4)
    g1, b1 = 1 + 0.2 * rnd(H, seed=7), 0.3 * rnd(H, seed=8)
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b = Sq, Bq, Nk, Bq, H, Bq, 1
    out, probs = torch.empty(Sq * Bq, H, device=DEV), torch.empty(Bq, Sq, Nk, device=DEV)
    qstats, ostats = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
    ln1 = torch.full((Sq * Bq, H), float("nan"), device=DEV)
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), gam.data_ptr(), bet.data_ptr()
    a.out, a.probs, a.qstats, a.out_stats = out.data_ptr(), probs.data_ptr(), qstats.data_ptr(), ostats.data_ptr()
    a.ln1_gamma, a.ln1_beta, a.ln1_out = g1.data_ptr(), b1.data_ptr(), ln1.data_ptr()
    prev = lib.dosx_attention_aligned_mode(0)
    try:
        o.attention_fwd(a)
    finally:
        lib.dosx_attention_aligned_mode(prev)
    assert err(ln1, torch.nn.functional.layer_norm(out.double(), (H,), g1.double(), b1.double(), 1e-5)) < 2e-5
    a.Nk = 330
    big_kv, big_p = rnd(330 * Bq, H, seed=2), torch.empty(Bq, Sq, 330, device=DEV)
    a.kvhat, a.probs = big_kv.data_ptr(), big_p.data_ptr()
    with pytest.raises(DosxError, match="ln1_out"):
        o.attention_fwd(a)
