"""Model-level parity on a real MI355X: the drop-in modules (libdosx programs through the C ABI)
against (a) the golden vectors generated from the reference and (b) the oracle evaluated live on the
same seeded inputs.  Tolerance on the predicted DOS vectors: 1e-4 RMSE (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from tests.util import batch_from, load, maxabs, rmse, sub

import ctypes as C
import math
import torch.nn.functional as F
from tests.gpu_util import (DEV, TOL, _FakeDist, _Hog, _attn_ref, _descs, _fat_crystals, _fatten, _graph, _mixed_jobs, _node_block, _philox_mask_numpy, _phonon, _random_crystals, _reduce, _ref, _scratch, _sliver_case, err, ops, prelu, rnd)  # noqa: F401
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


@pytest.mark.parametrize("B", [4, 64])
def test_against_oracle_live_with_split_bf16_feed_forward(monkeypatch, B):
    """the opt-in split-bf16 kernel under the plain feed-forward GEMMs of the Electron-DOS model (hidden 256:
    fc1 forward, fc2 forward, fc2's ReLU-masked input gradient - functional._FFN_BF16X3) against the oracle with EVERY tolerance of
    test_against_oracle_live unchanged (outputs, loss, every gradient's maximum / 99th percentile / median, one AdamW step)."""
    from dostransformer_amd import functional as Fn, ops as O_
    calls = []
    real = O_.gemm_bf16x3

    def counting(*a, **k):
        calls.append(k.get("w_layout", 0))
        return real(*a, **k)
    monkeypatch.setattr(Fn, "_FFN_BF16X3", True)
    monkeypatch.setattr(O_, "gemm_bf16x3", counting)
    test_against_oracle_live("edos", 256, 2, B)
    # three encoder stacks x 2 layers: fc1 + fc2 forward (+ the tail-row chains at 64 crystals), fc2's input gradient
    assert calls.count(0) >= 12 and calls.count(1) >= 6, calls


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


@pytest.mark.parametrize("H,T,B,steps,every", [(64, 1, 8, 200, 10), (128, 2, 64, 100, 10)])
def test_fp32_training_trajectory_tracks_fp64_oracle(H, T, B, steps, every):
    """The precision question of VERDICT r1: fp32 kernels against the reference's fp64 phonon arithmetic over a
    TRAJECTORY, not 1-3 steps.  BASELINE configs[0] (H64 T1 B8), 200 AdamW steps, and configs[1] (H128 T2 B64, the benchmark
    configuration), 100 steps (round 5 ran 200 there: 77 s of CPU oracle inside the GPU suite's wall time; round 4: 50).

    Three runs from the same initial weights on the same batches: the oracle in fp64 on the CPU (the reference's
    arithmetic, main_phDOS.py:15-16), the oracle in fp32 on the CPU (plain torch fp32: what `torch.float32` upstream
    would give), and the HIP path (fp32).  Checked at every `every`-th step on a held-out batch:
      * the loss curves of HIP-fp32 and fp64 agree within 1e-4 at EVERY step;
      * early on (<= 10 steps) the predicted DOS vectors agree within the north_star tolerance (1e-4 RMSE);
      * later the two fp32 runs both wander from the fp64 one — AdamW divides by sqrt(v): where a gradient element is at
        the fp32 noise floor its normalised update is O(lr) with a noise-determined sign, so ANY fp32 implementation
        decorrelates from fp64 on those elements at ~lr per step.  What is asserted is that the HIP path drifts no more
        than torch's own fp32 does (within 3x; measured on MI355X in round 2: see the printed line), i.e. the deviation
        is the arithmetic width, not the kernels.  DESIGN.md §4 carries the measured numbers."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    model = _phonon(H, T)
    p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    p32 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    n_b = 4
    b64 = [synth.phonon_batch(B, seed=500 + k, dtype=torch.float64) for k in range(n_b)]
    b32 = [synth.phonon_batch(B, seed=500 + k, dtype=torch.float32) for k in range(n_b)]
    gpu_b = [b.clone().to(DEV) for b in b32]
    probe64 = synth.phonon_batch(B, seed=599, dtype=torch.float64)
    probe32 = synth.phonon_batch(B, seed=599, dtype=torch.float32)
    probe_gpu = probe32.clone().to(DEV)
    tr = Trainer(model, lr=1e-4, beta=1.0, replay=True)
    s64, s32 = {}, {}
    hip_dos, cpu_dos, hip_loss, cpu_loss, early = 0.0, 0.0, 0.0, 0.0, 0.0
    torch.set_num_threads(8)
    for i in range(steps):
        lg = float(tr.step(gpu_b[i % n_b]))
        l64, _ = O.train_step("phonon", p64, s64, b64[i % n_b], 3, T, lr=1e-4, beta=1.0)
        l32, _ = O.train_step("phonon", p32, s32, b32[i % n_b], 3, T, lr=1e-4, beta=1.0)
        hip_loss = max(hip_loss, abs(lg - float(l64)))
        cpu_loss = max(cpu_loss, abs(float(l32) - float(l64)))
        if (i + 1) % every == 0:
            with torch.no_grad():
                dg, _, ds = model(probe_gpu)
                rg, _, rs = O.dostransformer_phonon_forward(p64, probe64, 3, T)
                cg, _, cs = O.dostransformer_phonon_forward(p32, probe32, 3, T)
            h = max(rmse(dg.cpu(), rg), rmse(ds.cpu(), rs))
            hip_dos = max(hip_dos, h)
            cpu_dos = max(cpu_dos, rmse(cg, rg), rmse(cs, rs))
            if i + 1 <= 10:
                early = h
    print(f"drift H{H} T{T} B{B}, {steps} AdamW steps vs the fp64 oracle: HIP fp32 DOS rmse {hip_dos:.3e} (at step 10: {early:.3e}), "
          f"torch-CPU fp32 DOS rmse {cpu_dos:.3e}; |loss diff| HIP {hip_loss:.3e}, torch-CPU fp32 {cpu_loss:.3e}")
    assert hip_loss < 1e-4, hip_loss
    assert early < 1e-4, early
    assert hip_dos < 3.0 * cpu_dos + 2e-5, (hip_dos, cpu_dos)


def test_model_attention_dropout_train_step_matches_oracle():
    """DOSTransformer_phonon(attn_drop=0.25): one training step through Trainer == the oracle with the same masks (loss and
    every gradient); replayed steps draw fresh masks each time (the seed lives on the device, outside the recording)."""
    from oracle import dos_oracle as O
    from dostransformer_amd import functional as Fn, synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 2, 118, 4, 32, DEV, 0.25)
    p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    g64 = synth.phonon_batch(5, seed=21, dtype=torch.float64)
    g = synth.phonon_batch(5, seed=21, dtype=torch.float32).to(DEV)
    tr = Trainer(model, lr=1e-4)
    Fn.DROP_MASK_LOG = []
    try:
        loss = tr.forward_backward(g)
        log = list(Fn.DROP_MASK_LOG)
    finally:
        Fn.DROP_MASK_LOG = None
    masks = {}
    for pre, t, m in log:
        masks.setdefault(pre, []).append(m.double().cpu())
    assert sorted(masks) == ["transformer", "transformer_self", "transformer_source"] and all(len(v) == 2 for v in masks.values())
    for k in p64:
        if p64[k].is_floating_point():
            p64[k].requires_grad_(True)
    dg, _, dsys = O.dostransformer_phonon_forward(p64, g64, 3, 2, drop_masks=masks)
    ref = O.loss_phonon(dg, dsys, g64.phdos, 1.0)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 2e-5
    fp = model.flat_params()
    for k, v in p64.items():
        if k in fp.G:
            e = float((fp.G[k].cpu().double() - v.grad).abs().max() / (v.grad.abs().max() + 1e-9))
            assert e < 3e-3, (k, e)
    # replay: every step must see a new mask (same batch, lr = 0 -> identical weights; the loss changes only through dropout)
    tr2 = Trainer(model, lr=0.0, weight_decay=0.0, replay=True)
    losses = [float(tr2.step(g)) for _ in range(4)]
    assert len(set(losses)) == 4, losses
    model.eval()
    with torch.no_grad():
        a = model(g)[0].clone()
        b = model(g)[0].clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_encoder_decoder_blocks_match_oracle(kind):
    """Encoder / Decoder called on their own (`DOSTransformer_phonon.py:126-145,174-183`, `DOSTransformer.py:100-122,151-161`)."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd._blocks import Decoder, Encoder
    torch.manual_seed(0)
    H = 32
    if kind == "phonon":
        enc, dec = Encoder(118, 4, H), Decoder(H, H)
        g = synth.phonon_batch(4, seed=7, dtype=torch.float32)
        ea = O.edge_features_sh1(g.edge_vec)
    else:
        enc, dec = Encoder(200, 41, H, n_global_feats=2), Decoder(2 * H, H)
        g = synth.edos_batch(3, seed=8, dtype=torch.float32)
        ea = g.edge_attr
    pe = {"GN_encoder." + k: v.detach().clone() for k, v in enc.state_dict().items()}
    pd = {"GN_decoder." + k: v.detach().clone() for k, v in dec.state_dict().items()}
    enc, dec = enc.to(DEV), dec.to(DEV)
    energies = torch.randn(51, H).to(DEV)
    args = (g.x.to(DEV), ea.to(DEV)) + ((g.glob.to(DEV),) if kind == "edos" else ()) + (g.batch.to(DEV), energies)
    outs = enc(*args)
    x_ref = O._mlp_prelu(pe, "GN_encoder.node_encoder", g.x)
    e_ref = O._mlp_prelu(pe, "GN_encoder.edge_encoder", ea)
    assert float((outs[0].cpu() - x_ref).abs().max()) < 2e-5 and float((outs[1].cpu() - e_ref).abs().max()) < 2e-5
    assert outs[-1].shape == (51, g.num_graphs, H) and torch.equal(outs[-1][:, 0], energies)
    xs = outs[0]
    if kind == "edos":
        u_ref = O._mlp_prelu(pe, "GN_encoder.global_encoder", g.glob.reshape(-1, 2))
        assert float((outs[2].cpu() - u_ref).abs().max()) < 2e-5
        y = dec(xs, outs[2], g.batch.to(DEV))
        y_ref = O._linear(pd, "GN_decoder.mlp.0", torch.cat([u_ref, O.scatter_sum(x_ref, g.batch, g.num_graphs)], 1))
    else:
        y = dec(xs, g.batch.to(DEV))
        y_ref = O._linear(pd, "GN_decoder.mlp.0", O.scatter_sum(x_ref, g.batch, g.num_graphs))
    assert float((y.cpu() - y_ref).abs().max()) < 5e-5
    y.sum().backward()
    assert enc.node_encoder[0].weight.grad is not None and dec.mlp[0].weight.grad is not None


def test_graphnetwork_prompt_branch():
    """Graphnetwork_phonon with 118 + H/2 wide node features takes `node_encoder_prompt` (`graphnetwork_phonon.py:150-153`);
    the plain `node_encoder` then stays without a gradient."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.graphnetwork_phonon import Graphnetwork_phonon
    torch.manual_seed(0)
    H = 32
    model = Graphnetwork_phonon(3, 118, 4, H, 51, DEV)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    g = synth.phonon_batch(4, seed=9, dtype=torch.float32)
    g.x = torch.cat([g.x, torch.randn(g.x.shape[0], H // 2, generator=torch.Generator().manual_seed(1))], 1)
    ref = O.graphnetwork_phonon_forward(p, g, 3)
    gg = g.clone().to(DEV)
    out = model(gg)
    assert rmse(out.detach().cpu(), ref) < 1e-4
    out.sum().backward()
    assert model.GN_encoder.node_encoder_prompt[0].weight.grad is not None
    assert model.GN_encoder.node_encoder[0].weight.grad is None
    pr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in p.items()}
    O.graphnetwork_phonon_forward(pr, g, 3).sum().backward()
    w = "GN_encoder.node_encoder_prompt.0.weight"
    gr = dict(model.named_parameters())[w].grad.cpu()
    assert float((gr - pr[w].grad).abs().max() / pr[w].grad.abs().max()) < 1e-3


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_models_with_overfull_nodes_match_the_oracle(kind):
    """Full models (mean aggregation: phonon, sum: eDOS) on a batch with 60- / 96- / 200-in-degree nodes against the oracle:
    outputs, loss, gradients; eager == replay bitwise; the device collate (Trainer.step_dataset: greedy tile table over the whole
    batch, round 6) gives bitwise the step on the host-collated batch; the crystal-aligned table of DeviceDataset.collate() the
    same gradients up to the grouping of three per-tile partial sums: an over-full node is cut the same way whatever shares the batch."""
    from oracle import dos_oracle as O
    from dostransformer_amd.batch import collate
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    B = 6
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        mk = lambda: DOSTransformer_phonon(3, 1, 118, 4, 64, DEV, 0.0)
        ref_dt, fwd = torch.float64, O.dostransformer_phonon_forward
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 1, 200, 41, 2, 64, DEV, 0.0)
        ref_dt, fwd = torch.float32, O.dostransformer_forward
    cs32 = _fat_crystals(kind, B, 9, torch.float32)
    cs_ref = _fat_crystals(kind, B, 9, ref_dt)
    g_ref, g = collate(cs_ref), collate(cs32)
    assert int((g.meta.seg_tile[2] != 0).sum()) >= 9
    model = mk()
    params = {k: (v.detach().clone().to(ref_dt) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    with torch.no_grad():
        rg, rx, rs = fwd(params, g_ref, 3, 1)
    tr = Trainer(model, lr=1e-3, beta=1.0)
    loss = tr.forward_backward(g.to(DEV))
    dg, xn, ds_ = tr.last_outputs
    rmse = lambda a, b: float(torch.sqrt(((a.double().cpu() - b.double()) ** 2).mean()))
    assert rmse(dg, rg) < 1e-4 and rmse(ds_, rs) < 1e-4 and rmse(xn, rx) < 1e-4 * max(1.0, float(rx.abs().max()))
    ref_loss, grads = O.train_step(kind, params, {}, g_ref, 3, 1, lr=1e-3, beta=1.0)
    assert abs(float(loss) - float(ref_loss)) < 2e-4
    fp = model.flat_params()
    for k, gr in grads.items():
        if gr is not None:
            e = float((fp.G[k].cpu().double() - gr.double()).abs().max() / (gr.abs().max() + 1e-6))
            assert e < 3e-3, (k, e)
    # eager, replay and device-collated replay: same trajectory, bit for bit
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    outs = []
    dsd = DeviceDataset(cs32, DEV)
    nmax = max(int(c["x"].shape[0]) for c in cs32)
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    gh = collate(cs32, n_max=nmax)
    gp = pad_batch(gh, *bucket_sizes(gh.meta.num_nodes, gh.meta.num_edges, 16, 256)).to(DEV)   # (eager on the same padded batch)
    for mode in ("eager", "replay", "dataset"):
        m2 = mk()
        m2.load_state_dict(sd0)
        m2 = m2.to(DEV)
        t2 = Trainer(m2, lr=1e-3, beta=1.0, replay=(mode != "eager"), bucket=(16, 256))
        for _ in range(3):
            if mode == "dataset":
                t2.step_dataset(dsd, list(range(B)), n_max=nmax)
            elif mode == "eager":
                t2.step(gp)
            else:
                t2.step(collate(cs32, n_max=nmax).to(DEV))
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().clone() for k, v in m2.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), ("eager vs replay", k)
        # host collate vs device collate straight into the bucket: round 6 - the device packs the node-aligned tiles greedily over
        # the whole batch like the host (csrc/csr.hip: collate_greedy_tiles_kernel), the same table tile for tile, so the two
        # trajectories are bitwise equal again (round 5's crystal-aligned device table grouped the EdgeModel's LayerNorm / PReLU
        # partial rows differently: equal to rounding only)
        assert torch.equal(outs[1][k], outs[2][k]), ("host collate vs device collate", k)
    # The crystal-aligned table still exists - DeviceDataset.collate() assembles a batch OBJECT from the per-crystal tilings on the
    # host - and is checked where the two tables can be told apart: the RAW gradients of one step.  Everything except the EdgeModel's
    # LayerNorm / PReLU parameter gradients (one partial row per TILE: another grouping of the same sums) is bitwise equal; those
    # three agree to fp32 rounding of their own scale - a dropped or doubled tile's partial row would be a 1e-2-level error
    grads = []
    for table in ("host", "crystal"):
        m3 = mk()
        m3.load_state_dict(sd0)
        m3 = m3.to(DEV)
        t3 = Trainer(m3, lr=1e-3, beta=1.0)
        t3.forward_backward(collate(cs32, n_max=nmax).to(DEV) if table == "host" else dsd.collate(list(range(B)), n_max=nmax))
        torch.cuda.synchronize()
        fp3 = m3.flat_params()
        grads.append({k: fp3.G[k].detach().cpu().clone() for k in fp3.names})
    per_tile = lambda k: "edge_mlp.1." in k or k.endswith("edge_mlp.2.weight")
    assert any(per_tile(k) for k in grads[0])
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        if per_tile(k):
            assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-3), ("per-tile partial rows", k)
        else:
            assert torch.equal(a, b), ("host table vs crystal-aligned table, raw gradient", k)


def test_shape_limits_are_explicit_errors():
    """The shape limits fail LOUDLY, with the limit in the message, and leave the library usable: the MFMA
    attention kernels stop at 256-wide rows (wider models take the unfused path, round 4: tests/test_gpu_models.py), and
    hidden > 512 exceeds the LayerNorm prologue of dosx_gemm.  The number of keys has no limit: more than 320 (the
    LDS-resident score row of the MFMA kernels) take the general kernels of csrc/attention_general.hip behind the same
    descriptor (tests/test_gpu_attention.py::test_attention_fwd_bwd)."""
    from dostransformer_amd._lib import DosxError, Attn
    o = ops()
    H, Sq, Bq = 32, 51, 2
    for Nk in (320, 321, 1000):                          # no limit on the keys: 321+ take the general kernels (same contract)
        x, kv = torch.randn(Sq * Bq, H, device=DEV), torch.randn(Nk * Bq, H, device=DEV)
        ones, zeros = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
        out, probs = torch.empty(Sq * Bq, H, device=DEV), torch.empty(Bq, Sq, Nk, device=DEV)
        a = Attn()
        a.Sq, a.Bq, a.Nk, a.Bk, a.H, a.q_stride_s, a.q_stride_b, a.flags = Sq, Bq, Nk, Bq, H, Bq, 1, 1 | 2
        a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kv.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        a.out, a.probs = out.data_ptr(), probs.data_ptr()
        o.attention_fwd(a)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all()) and float((probs.sum(-1) - 1).abs().max()) < 1e-5
    a.H = 260                                            # ... but the row kernels stop at 256 columns
    with pytest.raises(DosxError, match="H=260"):
        o.attention_fwd(a)
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    model = DOSTransformer_phonon(1, 1, 118, 4, 640, DEV, 0.0).to(DEV)
    g = synth.phonon_batch(2, seed=1, dtype=torch.float32).to(DEV)
    with pytest.raises(DosxError, match="hidden <= 512"):
        model(g)
    small = DOSTransformer_phonon(1, 1, 118, 4, 32, DEV, 0.0).to(DEV)
    assert bool(torch.isfinite(small(g)[0]).all())


@pytest.mark.parametrize("mode", ["eager", "replay"])
def test_crystal_with_more_than_320_atoms(mode):
    """A crystal of 330 atoms next to one of 5: the cross attention over atoms runs over 330 keys (zero-padded for the small
    crystal, `DOSTransformer_phonon.py:86-88`) - beyond the MFMA attention kernels, through csrc/attention_general.hip behind the
    same descriptor.  Outputs, loss and every gradient against the oracle in float64; replay = eager bitwise."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    H, T = 32, 2
    model = DOSTransformer_phonon(2, T, 118, 4, H, DEV, 0.0)
    params = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    gen = torch.Generator().manual_seed(4)
    cr = [synth.phonon_crystal(gen, n_atoms=330, n_out=4), synth.phonon_crystal(gen, n_atoms=5, n_out=4)]
    g_ref = collate(cr)
    g = collate([{k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in c.items()} for c in cr]).to(DEV)
    with torch.no_grad():
        rg, rx, rs = O.dostransformer_phonon_forward(params, g_ref, 2, T)
    tr = Trainer(model, lr=1e-4, beta=1.0, replay=(mode == "replay"))
    if mode == "replay":
        losses = [float(tr.step(g)) for _ in range(2)]     # the second step replays the recorded program
        model2 = DOSTransformer_phonon(2, T, 118, 4, H, DEV, 0.0)
        model2.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in params.items()})
        tr2 = Trainer(model2.to(DEV), lr=1e-4, beta=1.0)
        # (eager on the SAME ghost-padded batch the replayed bucket holds: the number of M-splits of a weight gradient
        #  depends on its row count, so the padding decides the summation order of the last bits)
        from dostransformer_amd.batch import bucket_sizes, pad_batch
        gp = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, *tr.bucket))
        eager = [float(tr2.step(gp)) for _ in range(2)]
        assert losses == eager
        for (k, v), (_, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
            assert torch.equal(v, v2), k
        return
    loss = tr.forward_backward(g)
    dg, xn, ds = tr.last_outputs
    rm = lambda a_, b_: float(torch.sqrt(((a_.double() - b_.double()) ** 2).mean()))
    assert rm(dg.cpu(), rg) < 1e-4 and rm(ds.cpu(), rs) < 1e-4
    ref_loss, grads = O.train_step("phonon", params, {}, g_ref, 2, T, lr=1e-4, beta=1.0)
    assert abs(float(loss) - float(ref_loss)) < 2e-4
    fp = model.flat_params()
    for k, gr in grads.items():
        if gr is None:
            assert k not in fp.G, k
        else:
            e = float((fp.G[k].cpu().double() - gr.double()).abs().max() / (gr.abs().max() + 1e-6))
            assert e < 2e-3, (k, e)


def test_standalone_modules_with_parameters_far_apart():
    """The fused feed-forward / NodeModel kernels reach both weight matrices of a layer through one 2 GiB buffer window.  A
    standalone module's parameters are separate torch allocations: put fc1 and fc2 (and the NodeModel's two Linears) at the two
    ends of a 3 GiB buffer - the modules pack them into one buffer for the call (functional.pack_params) and give the same
    bits as before."""
    from dostransformer_amd.layers import TransformerEncoder
    from dostransformer_amd._blocks import NodeModel
    torch.manual_seed(0)
    H = 64
    enc = TransformerEncoder(embed_dim=H, num_heads=1, layers=2).to(DEV)
    node = NodeModel(H).to(DEV)
    x = torch.randn(51, 3, H, device=DEV, requires_grad=True)
    kv = torch.randn(9, 3, H, device=DEV)
    xn = torch.randn(10, H, device=DEV, requires_grad=True)
    ei = torch.randint(0, 10, (2, 40), device=DEV)
    ea = torch.randn(40, H, device=DEV)

    def run():
        for t in (x, xn):
            t.grad = None
        enc.zero_grad(); node.zero_grad()
        y = enc(x, kv, kv)
        z = node(xn, ei, ea)
        (y.sum() + z.sum()).backward()
        return [y.detach().clone(), z.detach().clone(), x.grad.clone(), xn.grad.clone()] + \
            [p.grad.clone() for p in list(enc.parameters()) + list(node.parameters()) if p.grad is not None]
    ref = run()
    big = torch.empty(3 * 2 ** 30 // 4, device=DEV)
    with torch.no_grad():
        pairs = [(lay.fc1.weight, lay.fc2.weight) for lay in enc.layers] + [(node.node_mlp_2[0].weight, node.node_mlp_2[3].weight)]
        lo, hi = 0, big.numel()
        for w1, w2 in pairs:                              # first matrix from the front of the buffer, second from its end
            a_, b_ = big[lo:lo + w1.numel()].view_as(w1), big[hi - w2.numel():hi].view_as(w2)
            a_.copy_(w1); b_.copy_(w2)
            w1.data, w2.data = a_, b_
            lo, hi = lo + w1.numel(), hi - w2.numel()
    assert abs(enc.layers[0].fc1.weight.data_ptr() - enc.layers[0].fc2.weight.data_ptr()) > 2 ** 31
    far = run()
    assert len(ref) == len(far)
    for a_, b_ in zip(ref, far):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("kind,H", [("phonon", 384), ("edos", 384), ("phonon", 512)])
def test_models_with_hidden_beyond_256_match_the_oracle(kind, H):
    """DOSTransformer_phonon / DOSTransformer with hidden 384 and 512 (the unfused path: K != V encoder building blocks on
    the raw dense keys, row kernels for the 2H-wide LayerNorms of the GNN blocks) against the oracle: the three outputs, the
    loss, every live gradient; dead parameters stay dead; eager and replay give the same trajectory bit for bit."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    B, L, T = 5, 2, 2
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        mk = lambda: DOSTransformer_phonon(L, T, 118, 4, H, DEV, 0.0)
        ref_dt, fwd, cs_of = torch.float64, O.dostransformer_phonon_forward, synth.phonon_crystals
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(L, T, 200, 41, 2, H, DEV, 0.0)
        ref_dt, fwd, cs_of = torch.float32, O.dostransformer_forward, synth.edos_crystals
    g_ref, g = collate(cs_of(B, 11, ref_dt)), collate(cs_of(B, 11, torch.float32))
    model = mk()
    params = {k: (v.detach().clone().to(ref_dt) if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    with torch.no_grad():
        rg, rx, rs = fwd(params, g_ref, L, T)
    tr = Trainer(model, lr=1e-3, beta=1.0)
    loss = tr.forward_backward(g.to(DEV))
    dg, xn, ds_ = tr.last_outputs
    rmse = lambda a, b: float(torch.sqrt(((a.double().cpu() - b.double()) ** 2).mean()))
    assert rmse(dg, rg) < 1e-4 and rmse(ds_, rs) < 1e-4 and rmse(xn, rx) < 1e-4 * max(1.0, float(rx.abs().max()))
    clone = lambda d, dt=None: {k: (v.clone().to(dt) if (dt is not None and v.is_floating_point()) else v.clone()) for k, v in d.items()}
    ref_loss, grads = O.train_step(kind, clone(params), {}, g_ref, L, T, lr=1e-3, beta=1.0)      # (the step updates its params)
    assert abs(float(loss) - float(ref_loss)) < 2e-4
    fp = model.flat_params()
    errs = []
    for k, gr in grads.items():
        if gr is not None:
            assert k in fp.G, k
            e = float((fp.G[k].cpu().double() - gr.double()).abs().max() / (gr.abs().max() + 1e-6))
            errs.append((e, k, float(torch.quantile((fp.G[k].cpu().double() - gr.double()).abs().flatten()[:1000000], 0.99) / (gr.abs().max() + 1e-6))))
        else:
            assert k not in fp.G, k
    errs.sort(reverse=True)
    print(f"hidden {H} {kind}: largest gradient errors / tensor max (max, p99): " + ", ".join(f"{k} {e:.1e}/{q:.1e}" for e, k, q in errs[:4]))
    if kind == "phonon":            # the same step in plain torch fp32 on the CPU: how far fp32 itself is from the fp64 oracle
        _, g32 = O.train_step(kind, clone(params, torch.float32), {}, collate(cs_of(B, 11, torch.float32)), L, T, lr=1e-3, beta=1.0)
        e32 = sorted(((float((g32[k].double() - gr.double()).abs().max() / (gr.abs().max() + 1e-6)), k)
                      for k, gr in grads.items() if gr is not None), reverse=True)
        print(f"            torch-CPU fp32 against the same fp64 oracle: " + ", ".join(f"{k} {e:.1e}" for e, k in e32[:4]))
    # max: isolated activation-gate flips (a ReLU / PReLU input at fp32 resolution on one side of zero here, on the other in
    # the oracle) move single rows - DESIGN.md §4; the 99th percentile bounds the typical error
    #  (printed above for the phonon case: plain torch fp32 on the CPU shows maxima of the same size against the fp64 oracle)
    assert errs[0][0] < 3e-2, errs[:3]
    big = [t for t in errs if t[2] > 1e-3 and fp.G[t[1]].numel() > 100000]
    assert not big, big                     # a large tensor whose TYPICAL error is large would be a kernel bug
    assert sum(1 for t in errs if t[0] > 3e-3) <= 6, errs[:8]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    outs = []
    gd = g.to(DEV)
    for replay in (False, True):
        m2 = mk()
        m2.load_state_dict(sd0)
        m2 = m2.to(DEV)
        t2 = Trainer(m2, lr=1e-3, beta=1.0, replay=replay)
        from dostransformer_amd.batch import bucket_sizes, pad_batch
        gp = pad_batch(collate(cs_of(B, 11, torch.float32)), *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 8, 128)).to(DEV)
        for _ in range(3):
            t2.step(gp)
        torch.cuda.synchronize()
        outs.append({k: v.detach().cpu().clone() for k, v in m2.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), ("eager vs replay", k)


def test_graphnetwork_with_hidden_384_matches_the_oracle():
    """The GNN-only variant (graphnetwork_phonon.py:48-72) at hidden 384: outputs and gradients against the oracle."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.graphnetwork_phonon import Graphnetwork_phonon
    torch.manual_seed(0)
    H, L, B = 384, 2, 4
    model = Graphnetwork_phonon(L, 118, 4, H, 51, DEV)
    params = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    g64, g = collate(synth.phonon_crystals(B, 13, torch.float64)), collate(synth.phonon_crystals(B, 13, torch.float32)).to(DEV)
    out = model(g)
    p = {k: v.clone().requires_grad_(True) if v.is_floating_point() else v for k, v in params.items()}
    ref = O.graphnetwork_phonon_forward(p, g64, L)
    ref = ref[0] if isinstance(ref, tuple) else ref
    assert float(torch.sqrt(((out.detach().cpu().double() - ref.detach()) ** 2).mean())) < 1e-4
    w = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (ref * w).sum().backward()
    (out * w.float().to(DEV)).sum().backward()
    for k, v in model.named_parameters():
        rgd = p[k].grad
        if rgd is None:
            assert v.grad is None, k
            continue
        e = float((v.grad.cpu().double() - rgd).abs().max() / (rgd.abs().max() + 1e-6))
        assert e < 3e-3, (k, e)
