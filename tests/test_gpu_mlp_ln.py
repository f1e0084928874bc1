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
