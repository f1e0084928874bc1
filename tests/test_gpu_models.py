"""Model-level parity on a real MI355X: the drop-in modules (libdosx programs through the C ABI)
against (a) the golden vectors generated from the reference and (b) the oracle evaluated live on the
same seeded inputs.  Tolerance on the predicted DOS vectors: 1e-4 RMSE (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from tests.util import batch_from, load, maxabs, rmse, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"
DOS_RMSE = 1e-4          # north_star tolerance on the predicted DOS vector


def relerr(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_g1_multihead_attention():
    from dostransformer_amd.layers.multihead_attention import MultiheadAttention
    z = load("g1_mha.npz")
    mha = MultiheadAttention(16, 1).to(DEV)
    q = torch.from_numpy(z["f32/q"]).to(DEV).requires_grad_(True)
    kv = torch.from_numpy(z["f32/kv"]).to(DEV).requires_grad_(True)
    out = mha(q, kv, kv)
    assert maxabs(out.cpu(), z["f32/out"]) < 5e-6
    # gradient against torch autograd of the same expression
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    (out * w).sum().backward()
    q2 = q.detach().double().requires_grad_(True)
    kv2 = kv.detach().double().requires_grad_(True)
    a = torch.softmax(torch.bmm(q2.transpose(0, 1), kv2.permute(1, 2, 0)) * 16 ** -0.5, -1)
    (torch.bmm(a, kv2.transpose(0, 1)).transpose(0, 1) * w.double()).sum().backward()
    assert relerr(q.grad, q2.grad) < 5e-5 and relerr(kv.grad, kv2.grad) < 5e-5
    assert mha.in_proj_weight.grad is None and mha.out_proj.weight.grad is None
    # the float64 fixture (fp32 softmax quirk) is met within fp32 rounding as well
    assert maxabs(mha(torch.from_numpy(z["f64/q"]).to(DEV), *(torch.from_numpy(z["f64/kv"]).to(DEV),) * 2).cpu(),
                  z["f64/out"]) < 1e-5
    with pytest.raises(AssertionError):
        mha(q, kv[:3], kv)


@pytest.mark.parametrize("mode", ["cross", "self"])
def test_g2_transformer_encoder(mode):
    from dostransformer_amd.layers import TransformerEncoder
    z = load("g2_encoder.npz")
    enc = TransformerEncoder(embed_dim=16, num_heads=1, layers=2, attn_dropout=0.0)
    enc.load_state_dict(sub(z, "p/"))
    enc = enc.to(DEV)
    x = torch.from_numpy(z[f"{mode}/x"]).to(DEV).requires_grad_(True)
    if mode == "cross":
        kv = torch.from_numpy(z["cross/kv"]).to(DEV).requires_grad_(True)
        y = enc(x, kv, kv)
    else:
        y = enc(x, x, x)
    assert maxabs(y.cpu(), z[f"{mode}/y"]) < 2e-5
    (y * torch.from_numpy(z[f"{mode}/w"]).to(DEV)).sum().backward()
    assert relerr(x.grad, z[f"{mode}/dx"]) < 1e-4
    if mode == "cross":
        assert relerr(kv.grad, z["cross/dkv"]) < 1e-4
    dead = set(str(s) for s in z[f"{mode}/dead"])
    for k, p in enc.named_parameters():
        if k in dead:
            assert p.grad is None, k
        else:
            assert relerr(p.grad, z[f"{mode}/g/{k}"]) < 2e-4, k
    with pytest.raises(ValueError):
        enc(x)


def _load_model(model, z, prefix="p0/"):
    sd = {k: v.float() if v.is_floating_point() else v for k, v in sub(z, prefix).items()}
    model.load_state_dict(sd)
    return model.to(DEV)


def _check_full(z, model, kind, g, tol_grad=2e-3):
    from dostransformer_amd.train import Trainer
    model = _load_model(model, z)
    gd = g.to(DEV, dtype=torch.float32)
    dg, xn, ds = model(gd)
    assert rmse(dg.cpu(), z["dos_global"]) < DOS_RMSE and rmse(ds.cpu(), z["dos_system"]) < DOS_RMSE
    assert rmse(xn.cpu(), z["x_nodes"]) < DOS_RMSE
    # the reference caller's loss expression on our outputs (main_phDOS.py:109-114 / main_eDOS.py:111-123)
    if kind == "phonon":
        y = gd.phdos
        loss = torch.sqrt(torch.nn.functional.mse_loss(dg, y)) + torch.sqrt(torch.nn.functional.mse_loss(ds, y))
    else:
        y = torch.where(gd.y_ft < 0, torch.zeros_like(gd.y_ft), gd.y_ft).reshape(len(gd.mp_id), -1)
        loss = torch.sqrt(((y - dg) ** 2).mean(1)).mean() + torch.sqrt(((y - ds) ** 2).mean(1)).mean()
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)
    opt.zero_grad()
    loss.backward()
    dead = set(str(s) for s in z["dead_params"])
    worst = 0.0
    for k, p in model.named_parameters():
        if k in dead:
            assert p.grad is None, k
        else:
            assert p.grad is not None, k
            ref = torch.from_numpy(z["g/" + k])
            e = float((p.grad.cpu().double() - ref.double()).abs().max() / (ref.abs().max() + 1e-7))
            worst = max(worst, e)
            assert e < tol_grad, (k, e)
    # three optimizer steps exactly like the reference loop (torch.optim.AdamW on model.parameters())
    opt.step()
    p1 = sub(z, "p1/")
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert maxabs(v.cpu(), p1[k]) < 2e-6, ("p1", k)
    for _ in range(2):
        dg, _, ds = model(gd)
        if kind == "phonon":
            loss = torch.sqrt(torch.nn.functional.mse_loss(dg, y)) + torch.sqrt(torch.nn.functional.mse_loss(ds, y))
        else:
            loss = torch.sqrt(((y - dg) ** 2).mean(1)).mean() + torch.sqrt(((y - ds) ** 2).mean(1)).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    p3 = sub(z, "p3/")
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert maxabs(v.cpu(), p3[k]) < 5e-6, ("p3", k)
    for k in dead:
        assert torch.equal(model.state_dict()[k].cpu(), sub(z, "p0/")[k].float())
    return worst


def test_g5_phonon_full():
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    z = load("g5_phonon.npz")
    _check_full(z, DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0), "phonon", batch_from(z))


def test_g6_edos_full():
    from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
    z = load("g6_edos.npz")
    _check_full(z, DOSTransformer(3, 2, 200, 41, 2, 16, DEV, 0.0), "edos", batch_from(z))


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_fused_trainer_matches_golden_trajectory(kind):
    from dostransformer_amd.train import Trainer
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        z = load("g5_phonon.npz")
        model = _load_model(DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0), z)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        z = load("g6_edos.npz")
        model = _load_model(DOSTransformer(3, 2, 200, 41, 2, 16, DEV, 0.0), z)
    g = batch_from(z).to(DEV, dtype=torch.float32)
    tr = Trainer(model, lr=1e-4, beta=1.0)
    loss = tr.step(g)
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    p1 = sub(z, "p1/")
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert maxabs(v.cpu(), p1[k]) < 2e-6, ("p1", k)
    tr.step(g)
    tr.step(g)
    p3 = sub(z, "p3/")
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert maxabs(v.cpu(), p3[k]) < 5e-6, ("p3", k)


def test_g7_batch_composition():
    from dostransformer_amd.batch import graph_meta
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    z = load("g7_batch_composition.npz")
    model = _load_model(DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0), z).eval()
    a = batch_from(z, "alone/b/").to(DEV, dtype=torch.float32)
    b = batch_from(z, "both/b/").to(DEV, dtype=torch.float32)
    with torch.no_grad():
        oa, ob = model(a), model(b)
    assert rmse(oa[0].cpu(), z["alone/dos_global"]) < DOS_RMSE and rmse(oa[2].cpu(), z["alone/dos_system"]) < DOS_RMSE
    assert rmse(ob[0].cpu(), z["both/dos_global"]) < DOS_RMSE and rmse(ob[2].cpu(), z["both/dos_system"]) < DOS_RMSE
    assert maxabs(oa[0][0].cpu(), ob[0][0].cpu()) > 1e-3          # unmasked padding: batch mates matter
    a2 = batch_from(z, "alone/b/").to(DEV, dtype=torch.float32)
    graph_meta(a2, DEV, n_max=11)                                  # pad to the global Nmax -> same as batched
    with torch.no_grad():
        oa2 = model(a2)
    assert maxabs(oa2[0][0].cpu(), ob[0][0].cpu()) < 1e-5


def test_g8_graphnetworks():
    from dostransformer_amd.embedder_eDOS.graphnetwork import Graphnetwork
    from dostransformer_amd.embedder_phDOS.graphnetwork_phonon import Graphnetwork_phonon
    z = load("g8_graphnetwork_phonon.npz")
    model = _load_model(Graphnetwork_phonon(3, 118, 4, 16, 51, DEV), z)
    dos = model(batch_from(z).to(DEV, dtype=torch.float32))
    assert rmse(dos.cpu(), z["dos"]) < DOS_RMSE
    (dos * torch.from_numpy(z["w"]).float().to(DEV)).sum().backward()
    dead = set(str(s) for s in z["dead_params"])
    for k, p in model.named_parameters():
        if k in dead:
            assert p.grad is None, k
        else:
            assert relerr(p.grad, z["g/" + k]) < 2e-3, k
    z = load("g8_graphnetwork_edos.npz")
    model = _load_model(Graphnetwork(3, 200, 41, 2, 16, 201, DEV), z)
    dos, xn = model(batch_from(z).to(DEV, dtype=torch.float32))
    assert rmse(dos.cpu(), z["dos"]) < DOS_RMSE and rmse(xn.cpu(), z["x_nodes"]) < DOS_RMSE
    (dos * torch.from_numpy(z["w"]).to(DEV)).sum().backward()
    dead = set(str(s) for s in z["dead_params"])
    for k, p in model.named_parameters():
        if k in dead:
            assert p.grad is None, k
        else:
            assert relerr(p.grad, z["g/" + k]) < 2e-3, k


# per-tensor gradient tolerance (max abs error / max abs of the tensor) = 10 x the error observed on MI355X (round 2,
# printed by the test); phonon: fp32 kernels against the fp64 oracle, eDOS: against the fp32 oracle (torch CPU)
GRAD_TOL = {("phonon", 64, 1, 8): 3e-3, ("phonon", 128, 2, 16): 3e-3, ("edos", 64, 2, 6): 2e-2, ("edos", 256, 2, 4): 2e-2,
            ("phonon", 128, 2, 64): 3e-3, ("edos", 256, 2, 64): 2e-2, ("edos", 256, 4, 32): 2e-2}


# typical element error of a gradient tensor, relative to the tensor maximum: ~3-4 x the largest values observed on MI355X
# over the seven cases (99th percentile 3.1e-4 - embeddings.weight, phonon H128 B64: one gate flip moves a whole 128-wide row
# of a 51-row tensor -, median 2.2e-5); a uniform 1e-3 error of the kernels fails the median bound by a factor of ten
GRAD_TOL_P99 = {"phonon": 1e-3, "edos": 1e-3}
GRAD_TOL_MEDIAN = {"phonon": 1e-4, "edos": 1e-4}


@pytest.mark.parametrize("kind,H,T,B", [("phonon", 64, 1, 8), ("phonon", 128, 2, 16), ("edos", 64, 2, 6),
                                         ("edos", 256, 2, 4),
                                         # BASELINE.json configs[1], [2] and the per-GPU shard of [4] at FULL size
                                         ("phonon", 128, 2, 64), ("edos", 256, 2, 64), ("edos", 256, 4, 32)])
def test_against_oracle_live(kind, H, T, B):
    """BASELINE configs at oracle-feasible batch sizes: outputs, loss, every gradient and one AdamW
    step against the oracle (fp64 for phonon like main_phDOS.py:15-16, fp32 for eDOS)."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        model = DOSTransformer_phonon(3, T, 118, 4, H, DEV, 0.0)
        g_ref = synth.phonon_batch(B, seed=11, dtype=torch.float64)
        g = synth.phonon_batch(B, seed=11, dtype=torch.float32)
        ref_dt = torch.float64
        fwd = O.dostransformer_phonon_forward
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        model = DOSTransformer(3, T, 200, 41, 2, H, DEV, 0.0)
        g_ref = synth.edos_batch(B, seed=12, dtype=torch.float32)
        g = synth.edos_batch(B, seed=12, dtype=torch.float32)
        ref_dt = torch.float32
        fwd = O.dostransformer_forward
    params = {k: (v.detach().clone().to(ref_dt) if v.is_floating_point() else v.clone())
              for k, v in model.state_dict().items()}
    model = model.to(DEV)
    g = g.to(DEV)
    with torch.no_grad():
        rg, rx, rs = fwd(params, g_ref, 3, T)
    tr = Trainer(model, lr=1e-4, beta=1.0)
    loss = tr.forward_backward(g)
    dg, xn, ds = tr.last_outputs
    assert rmse(dg.cpu(), rg) < DOS_RMSE and rmse(ds.cpu(), rs) < DOS_RMSE
    assert rmse(xn.cpu(), rx) < DOS_RMSE * max(1.0, float(rx.abs().max()))
    state = {}
    ref_loss, grads = O.train_step(kind, params, state, g_ref, 3, T, lr=1e-4, beta=1.0)
    assert abs(float(loss) - float(ref_loss)) < 2e-4
    fp = model.flat_params()
    tol = GRAD_TOL[(kind, H, T, B)]
    worst = (0.0, None)
    worst99 = (0.0, None)          # TYPICAL error: the 99th percentile of the element errors of a tensor (tensors of >= 256
    worst50 = (0.0, None)          # elements), and the median - a handful of activation-gate flips may reach
    for k, gr in grads.items():    # `tol`, a uniform regression of the kernels cannot hide under it
        if gr is None:
            assert k not in fp.G, k
        else:
            err = (fp.G[k].cpu().double() - gr.double()).abs().reshape(-1) / (gr.abs().max() + 1e-6)
            worst = max(worst, (float(err.max()), k))
            if err.numel() >= 256:
                worst99 = max(worst99, (float(torch.quantile(err[:1 << 24], 0.99)), k))
                worst50 = max(worst50, (float(err.median()), k))
    print(f"oracle-live {kind} H{H} T{T} B{B}: worst per-tensor gradient error (relative to the tensor max) {worst[0]:.3e} at {worst[1]}"
          f"; worst 99th percentile {worst99[0]:.3e} at {worst99[1]}; worst median {worst50[0]:.3e} at {worst50[1]}")
    assert worst[0] < tol, worst
    assert worst99[0] < GRAD_TOL_P99[kind], worst99
    assert worst50[0] < GRAD_TOL_MEDIAN[kind], worst50
    tr.optimizer_step()
    # Adam's first step moves every element by ~lr*sign(g): where |g| is at the noise floor of the fp32
    # (GPU) vs fp64 (oracle) gradient the sign itself is ill-conditioned, so compare the update only
    # where the gradient is resolved, and bound it by 2*lr everywhere.
    for k, v in model.state_dict().items():
        if not v.is_floating_point():
            continue
        d = (v.cpu().double() - params[k].double()).abs()
        assert float(d.max()) <= 2.1e-4, k
        gr = grads.get(k)
        if gr is not None:
            ok = gr.abs() >= 1e-2 * gr.abs().max()
            assert float(d[ok].max()) < 5e-6, k
        else:
            assert float(d.max()) == 0.0, k


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_ghost_padding_is_exact(kind):
    """pad_batch (shape buckets for HIP-graph replay) must not change outputs or gradients."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        model = DOSTransformer_phonon(3, 2, 118, 4, 64, DEV, 0.0).to(DEV)
        g = synth.phonon_batch(7, seed=31, dtype=torch.float32)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        model = DOSTransformer(3, 1, 200, 41, 2, 64, DEV, 0.0).to(DEV)
        g = synth.edos_batch(5, seed=32, dtype=torch.float32)
    n_pad, e_pad = bucket_sizes(g.meta.num_nodes, g.meta.num_edges)
    gp = pad_batch(g, n_pad + 64, e_pad + 512).to(DEV)
    g = g.to(DEV)
    tr = Trainer(model)
    l0 = tr.forward_backward(g)
    o0 = [t.clone() for t in tr.last_outputs]
    g0 = model.flat_params().grad.clone()
    l1 = tr.forward_backward(gp)
    o1 = tr.last_outputs
    g1 = model.flat_params().grad
    N = g.meta.num_nodes
    assert torch.equal(o0[0], o1[0]) and torch.equal(o0[2], o1[2]) and torch.equal(o0[1], o1[1][:N])
    assert float(l0) == float(l1)
    assert float((g0 - g1).abs().max()) <= 1e-6 * float(g0.abs().max())      # slab split points move, math does not


@pytest.mark.parametrize("mode", ["graph", "replay"])
@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_graph_replay_matches_eager(kind, mode):
    """Trainer(graph=True) / Trainer(replay=True): captured HIP graphs / recorded launch lists per shape bucket (two
    streams in replay mode) reproduce the eager trajectory on the same ghost-padded batches BITWISE (no atomics anywhere,
    fixed summation orders), and the eager trajectory on the un-padded batches up to the noise Adam makes of the moved
    slab split points (padding changes M of the weight-gradient kernels, i.e. fp32 summation order at the 1e-7 level)."""
    import copy
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        mk = lambda: DOSTransformer_phonon(3, 2, 118, 4, 64, DEV, 0.0)
        batches = [synth.phonon_batch(6, seed=40 + k, dtype=torch.float32).to(DEV) for k in range(3)]
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 1, 200, 41, 2, 64, DEV, 0.0)
        batches = [synth.edos_batch(4, seed=50 + k, dtype=torch.float32).to(DEV) for k in range(3)]
    padded = [pad_batch(b, *bucket_sizes(b.meta.num_nodes, b.meta.num_edges)) for b in batches]
    torch.manual_seed(1)
    m_e = mk().to(DEV)
    m_p, m_g = mk(), mk()
    m_p.load_state_dict(copy.deepcopy(m_e.state_dict()))
    m_g.load_state_dict(copy.deepcopy(m_e.state_dict()))
    m_p, m_g = m_p.to(DEV), m_g.to(DEV)
    te, tp = Trainer(m_e, lr=1e-3), Trainer(m_p, lr=1e-3)
    tg = Trainer(m_g, lr=1e-3, graph=(mode == "graph"), replay=(mode == "replay"))
    for i in range(7):                      # revisits buckets -> replays, not only captures
        le = te.step(batches[i % 3])
        lp = tp.step(padded[i % 3])
        lg = tg.step(batches[i % 3])        # pads on the fly
        assert float(lp) == float(lg), i
        assert abs(float(le) - float(lg)) < 1e-5 * max(1.0, abs(float(le)))
    assert len(tg._slots) == 3 and tg.slot_hits == 4
    for (k, a), (_, b), (_, c) in zip(m_e.state_dict().items(), m_p.state_dict().items(), m_g.state_dict().items()):
        if a.is_floating_point():
            assert torch.equal(b, c), k
            assert float((a - c).abs().max()) < 7e-3, k          # 7 steps of lr 1e-3 bound any element's drift


def test_g9_eval_loops_match_reference():
    """`dostransformer_amd.evaluate.test / test_phonon` against the reference's own utils.test / test_phonon."""
    from dostransformer_amd import evaluate
    from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    z = load("g9_eval.npz")
    model = DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0)
    sd = {k: (v.float() if v.is_floating_point() else v) for k, v in sub(z, "ph/p0/").items()}
    model.load_state_dict(sd)
    model = model.to(DEV)
    loader = [batch_from(z, "ph/b0/").to(DEV, dtype=torch.float32), batch_from(z, "ph/b1/").to(DEV, dtype=torch.float32)]
    m = evaluate.test_phonon(model, loader, torch.nn.L1Loss(), evaluate.r2, DEV)
    assert np.allclose(m, z["ph/metrics"], rtol=2e-5, atol=2e-6), (m, z["ph/metrics"])
    assert not model.training                      # utils.py:118: model.eval()
    model = DOSTransformer(3, 2, 200, 41, 2, 16, DEV, 0.0)
    model.load_state_dict(sub(z, "e/p0/"))
    model = model.to(DEV)
    loader = [batch_from(z, "e/b0/").to(DEV), batch_from(z, "e/b1/").to(DEV)]
    rmse_, mse_, mae_, r2_, preds_y = evaluate.test(model, loader, torch.nn.L1Loss(), evaluate.r2, DEV)
    assert np.allclose([rmse_, mse_, mae_, r2_], z["e/metrics"], rtol=5e-5, atol=5e-6)
    ids, preds, y, emb = preds_y[0]
    assert ids == [str(s) for s in z["e/mp_id"]]
    assert maxabs(preds, z["e/preds"]) < 2e-5 and maxabs(y, z["e/y"]) == 0.0 and maxabs(emb, z["e/embeddings"]) < 2e-4


def test_checkpoint_resume_is_exact(tmp_path):
    """2 steps + save + load into a fresh model/trainer + 1 step == 3 uninterrupted steps (bitwise)."""
    from dostransformer_amd import checkpoint, synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    g = synth.phonon_batch(4, seed=3, dtype=torch.float32).to(DEV)
    torch.manual_seed(0)
    a = DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0).to(DEV)
    ta = Trainer(a, lr=1e-3)
    for _ in range(2):
        ta.step(g)
    path = str(tmp_path / "ck.pt")
    checkpoint.save(path, a, ta, extra={"epoch": 7})
    ta.step(g)
    torch.manual_seed(123)                               # different init: everything must come from the file
    b = DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0).to(DEV)
    tb = Trainer(b, lr=5e-2)
    assert checkpoint.load(path, b, tb) == {"epoch": 7}
    assert tb.step_count == 2 and tb.lr == 1e-3
    tb.step(g)
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va.cpu(), vb.cpu()), k


def test_foreign_device_batch_uses_device_csr():
    """A PyG-style batch that is already on the GPU (unsorted edges, no metadata) goes through dosx_csr_build and
    gives the same outputs as the host-collated, destination-sorted batch."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import CrystalBatch
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 1, 118, 4, 16, DEV, 0.0).to(DEV)
    g = synth.phonon_batch(5, seed=11, dtype=torch.float32, sort_edges=False)
    perm = torch.randperm(g.edge_index.shape[1], generator=torch.Generator().manual_seed(5))
    foreign = CrystalBatch({"x": g.x.to(DEV), "edge_index": g.edge_index[:, perm].to(DEV), "edge_vec": g.edge_vec[perm].to(DEV),
                            "batch": g.batch.to(DEV), "system": g.system.to(DEV), "phdos": g.phdos.to(DEV)}, 5, meta=None)
    ref = synth.phonon_batch(5, seed=11, dtype=torch.float32).to(DEV)
    with torch.no_grad():
        a = model(foreign)
        b = model(ref)
    assert foreign.meta is not None and foreign.meta.src.is_cuda and foreign.meta.edge_perm is not None
    for u, v in zip(a, b):
        assert maxabs(u.cpu(), v.cpu()) < 2e-6


@pytest.mark.parametrize("mode", ["eager", "replay"])
def test_data_parallel_buckets_match_single_process(mode):
    """World-size-1 RCCL group: the bucketed data-parallel step (early bucket all-reduced under the GNN backward,
    GNN bucket at the end) must leave exactly the parameters of the plain step."""
    import os
    import torch.distributed as td
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    from dostransformer_amd.dist import DataParallel
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    created = not td.is_initialized()
    if created:
        td.init_process_group("nccl", rank=0, world_size=1)
    try:
        gs = []
        for k in range(2):
            g = synth.phonon_batch(6, seed=20 + k, dtype=torch.float32)
            gs.append(pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges)).to(DEV))
        out = []
        for dist in (None, DataParallel()):
            torch.manual_seed(0)
            model = DOSTransformer_phonon(3, 1, 118, 4, 32, DEV, 0.0).to(DEV)
            tr = Trainer(model, lr=1e-3, dist=dist, replay=(mode == "replay"))
            for i in range(4):
                tr.step(gs[i % 2], 6)
            torch.cuda.synchronize()
            out.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
            if dist is not None:
                fp = model.flat_params()
                assert 0 < fp.n_last < fp.n_late < fp.total and tr._early_work is None and tr._mid_work is None
        for k in out[0]:
            assert torch.equal(out[0][k], out[1][k]), k
    finally:
        if created:
            td.destroy_process_group()


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_device_collate_matches_host(kind):
    """loader.DeviceDataset.collate (dosx_collate + row gathers, all on the GPU) == batch.collate on the host."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.loader import DeviceDataset
    cs = synth.phonon_crystals(12, seed=31, dtype=torch.float32) if kind == "phonon" else synth.edos_crystals(12, seed=32, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    for sel in ([3, 0, 7], [11], list(range(12)), [5, 5, 2]):
        d = ds.collate(sel)
        h = collate([cs[i] for i in sel])
        torch.cuda.synchronize()
        for k in h.keys():
            if isinstance(h[k], torch.Tensor):
                assert torch.equal(d[k].cpu(), h[k]), k
            else:
                assert d[k] == h[k], k
        assert (d.meta.num_nodes, d.meta.num_edges, d.meta.num_graphs, d.meta.n_max) == \
               (h.meta.num_nodes, h.meta.num_edges, h.meta.num_graphs, h.meta.n_max)
        for k in ("src", "dst", "rowptr_dst", "perm_src", "rowptr_src", "graph_ptr", "node_graph", "dense_row", "inv_deg"):
            assert torch.equal(getattr(d.meta, k).cpu(), getattr(h.meta, k)), k
        assert d.meta.edge_perm is None


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_predictor_replay_is_bitwise_eager(kind):
    """predict.Predictor (recorded forward program per shape bucket, ghost padded) == model(batch) under no_grad,
    for single crystals and batches, on first use of a bucket (record) and on revisits (replay)."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.predict import Predictor
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        model = DOSTransformer_phonon(3, 2, 118, 4, 64, DEV, 0.0).to(DEV)
        cs = synth.phonon_crystals(10, seed=61, dtype=torch.float32)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        model = DOSTransformer(3, 1, 200, 41, 2, 64, DEV, 0.0).to(DEV)
        cs = synth.edos_crystals(10, seed=62, dtype=torch.float32)
    model.eval()
    pred = Predictor(model)
    sels = [[0], [1], [0], [2, 3, 4], [9], [2, 3, 4], [1], list(range(10)), [4, 3, 2]]
    for sel in sels:
        g = collate([cs[i] for i in sel]).to(DEV)
        with torch.no_grad():
            ref = [t.clone() for t in model(g)]
        out = pred(g)
        torch.cuda.synchronize()
        for a, b in zip(ref, out):
            assert a.shape == b.shape and torch.equal(a, b), sel
    assert 1 <= len(pred._slots) < len(sels)          # buckets were revisited, i.e. replays happened


def test_eval_loop_accepts_predictor():
    from dostransformer_amd import evaluate, synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.predict import Predictor
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 2, 118, 4, 64, DEV, 0.0).to(DEV)
    cs = synth.phonon_crystals(12, seed=63, dtype=torch.float32)
    loader = [collate(cs[i:i + 4]).to(DEV) for i in (0, 4, 8, 0)]
    a = evaluate.test_phonon(model, loader)
    b = evaluate.test_phonon(Predictor(model), loader)
    assert a == b


def test_example_driver_end_to_end(tmp_path):
    """examples/train_phonon.py: structures -> GPU neighbour list -> DeviceDataset -> replayed Trainer -> replayed
    eval -> checkpoint; the loss must go down and the checkpoint must reload into a fresh module."""
    import importlib.util
    import os
    from dostransformer_amd import checkpoint
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_phonon", os.path.join(root, "examples", "train_phonon.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "best.pt")
    res = mod.main(["--epochs", "8", "--crystals", "120", "--hidden", "32", "--transformer", "1", "--batch-size", "16",
                    "--lr", "2e-3", "--out", out])
    h = res["train_loss"]
    assert len(h) == 8 and all(np.isfinite(h)) and h[-1] < 0.8 * h[0], h
    assert np.isfinite(res["best_valid_rmse"]) and os.path.exists(out)
    fresh = DOSTransformer_phonon(3, 1, 118, 4, 32, DEV, 0.0).to(DEV)
    checkpoint.load(out, fresh)
