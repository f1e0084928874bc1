"""Round-2 GPU regressions: fp64 batches through the replay paths, optimizer state across re-homing, and the
fp32-for-fp64 precision question on a TRAINING TRAJECTORY (the phonon reference runs in fp64, `main_phDOS.py:15-16`)."""
import copy

import numpy as np
import pytest
import torch

from tests.util import rmse

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _phonon(H=64, T=2):
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    return DOSTransformer_phonon(3, T, 118, 4, H, DEV, 0.0)


@pytest.mark.parametrize("mode", ["graph", "replay"])
def test_replay_takes_fp64_batches(mode):
    """float64 is the phonon pipeline's dtype (main_phDOS.py:15-16; synth.phonon_batch, DeviceDataset default).  Several
    fp64 batches of ONE shape bucket: every replayed step must read the batch it was given (round-1 bug: a cast made
    while recording was not part of the recorded program, later steps kept reading the first batch's fp32 copy)."""
    from dostransformer_amd import synth
    from dostransformer_amd.train import Trainer
    n_atoms = [3, 5, 2, 7, 4, 6]
    batches64 = [synth.phonon_batch(6, seed=70 + k, dtype=torch.float64, n_atoms=n_atoms).to(DEV) for k in range(3)]
    batches32 = [synth.phonon_batch(6, seed=70 + k, dtype=torch.float32, n_atoms=n_atoms).to(DEV) for k in range(3)]
    assert batches64[0].x.dtype == torch.float64 and batches64[0].edge_vec.dtype == torch.float64
    torch.manual_seed(1)
    m_e = _phonon().to(DEV)
    m_r = _phonon()
    m_r.load_state_dict(copy.deepcopy(m_e.state_dict()))
    m_r = m_r.to(DEV)
    te = Trainer(m_e, lr=1e-3)
    tr = Trainer(m_r, lr=1e-3, graph=(mode == "graph"), replay=(mode == "replay"))
    for i in range(6):
        le = te.step(batches32[i % 3])
        lr_ = tr.step(batches64[i % 3])
        # fp64 -> fp32 conversion of the inputs is the same rounding synth applies for the fp32 batches
        assert abs(float(le) - float(lr_)) < 1e-5 * max(1.0, abs(float(le))), i
    assert len(tr._slots) == 1 and tr.slot_hits == 5
    for (k, a), (_, b) in zip(m_e.state_dict().items(), m_r.state_dict().items()):
        if a.is_floating_point():
            assert float((a - b).abs().max()) < 2e-5, k


def test_predictor_takes_fp64_batches():
    from dostransformer_amd import synth
    from dostransformer_amd.predict import Predictor
    torch.manual_seed(0)
    model = _phonon().to(DEV).eval()
    pred = Predictor(model)
    n_atoms = [4, 4, 9]
    for k in range(4):
        g64 = synth.phonon_batch(3, seed=80 + k, dtype=torch.float64, n_atoms=n_atoms).to(DEV)
        g32 = synth.phonon_batch(3, seed=80 + k, dtype=torch.float32, n_atoms=n_atoms).to(DEV)
        with torch.no_grad():
            ref = [t.clone() for t in model(g32)]
        out = pred(g64)
        torch.cuda.synchronize()
        for a, b in zip(ref, out):
            assert torch.equal(a, b), k
    assert len(pred._slots) == 1


def test_predictor_cache_follows_field_assignment():
    """the ghost-padded copy cached on the batch must not survive `g.x = ...` / `g['edge_vec'] = ...`."""
    from dostransformer_amd import synth
    from dostransformer_amd.predict import Predictor
    torch.manual_seed(0)
    model = _phonon().to(DEV).eval()
    pred = Predictor(model)
    g = synth.phonon_batch(3, seed=90, dtype=torch.float32).to(DEV)
    a = [t.clone() for t in pred(g)]
    g.x = g.x * 0.5
    g["edge_vec"] = g.edge_vec * 0.9
    b = [t.clone() for t in pred(g)]
    with torch.no_grad():
        ref = model(g)
    assert not torch.equal(a[0], b[0])
    for u, v in zip(ref, b):
        assert torch.equal(u, v)
    g.to(DEV)                       # nothing moves: the cached padded copy stays
    assert getattr(g, "_dosx_padded", None) is not None


def test_optimizer_state_survives_cpu_load_then_to_gpu(tmp_path):
    """checkpoint.load(path, model, trainer) while the model is still on the CPU, then model.to('cuda'): the AdamW moments
    must follow the re-homed parameters (they were silently zeroed while step_count kept counting)."""
    from dostransformer_amd import checkpoint, synth
    from dostransformer_amd.train import Trainer
    gs = [synth.phonon_batch(5, seed=100 + k, dtype=torch.float32).to(DEV) for k in range(3)]
    torch.manual_seed(3)
    m0 = _phonon(32, 1).to(DEV)
    t0 = Trainer(m0, lr=1e-3)
    for g in gs[:2]:
        t0.step(g)
    path = str(tmp_path / "ck.pt")
    checkpoint.save(path, m0, t0)
    t0.step(gs[2])                                    # the continuation to reproduce
    torch.manual_seed(99)
    m1 = _phonon(32, 1)                               # on the CPU
    t1 = Trainer(m1, lr=1e-3)
    checkpoint.load(path, m1, t1)                     # weights_only=True inside
    assert t1.step_count == 2
    m1 = m1.to(DEV)
    t1.step(gs[2])
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), k
    assert float(t1._m.abs().max()) > 0


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


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_collate_into_matches_pad_batch(kind):
    """loader.DeviceDataset.collate_into (dosx_collate_padded: selection + feature gathers + ghost tail straight into a
    bucket's static buffers) == pad_batch(collate(...)) on every field and index array the kernels read, bit for bit."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import _Slot, _META_TENSORS
    cs = synth.phonon_crystals(14, seed=41, dtype=torch.float32) if kind == "phonon" else synth.edos_crystals(14, seed=42, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    for sel in ([3, 0, 7], [11], list(range(14)), [5, 5, 2, 13]):
        idx, N, E, n_max = ds.bucket_dims(sel, n_max=45)
        n_pad, e_pad = bucket_sizes(N, E, 16, 256)
        t = ds._f32_tables()
        slot = _Slot.empty(kind, DEV, len(sel), n_pad, e_pad, n_max, t["x"].shape[1], t["edge"].shape[1], t["target"].shape[1],
                           tiled=True)
        for v in list(slot.g._fields.values()) + [getattr(slot.g.meta, k) for k in _META_TENSORS]:
            if torch.is_tensor(v):
                v.fill_(77)                                  # stale contents of a previous batch must all be overwritten
        ds.collate_into(slot.g, idx, slot.scratch)
        ref = pad_batch(collate([cs[i] for i in sel], n_max=45), n_pad, e_pad)
        torch.cuda.synchronize()
        for k in slot.fields:
            a, b = slot.g[k].cpu(), ref[k]
            assert torch.equal(a.reshape(-1), b.to(a.dtype).reshape(-1)), (k, sel)
        for k in _META_TENSORS:
            assert torch.equal(getattr(slot.g.meta, k).cpu(), getattr(ref.meta, k)), (k, sel)
        assert (slot.g.meta.num_nodes, slot.g.meta.num_edges, slot.g.meta.n_max) == (ref.meta.num_nodes, ref.meta.num_edges, 45)
        # node-aligned row tiles of the message GEMM: the device tiling is crystal-aligned, the host one greedy over the whole
        # batch - different tables, both valid: monotone, <= 48 rows, tile edges = CSR pointers of its node range, full cover
        from dostransformer_amd.batch import SEG_TILE_ROWS
        for tt, nreal in ((slot.g.meta.seg_tile.cpu().numpy(), N), (ref.meta.seg_tile.numpy(), N)):
            eb, nb = tt[0], tt[1]
            rp = ref.meta.rowptr_dst.numpy()
            assert eb[0] == 0 and nb[0] == 0 and eb[-1] == e_pad and nb[-1] == n_pad
            assert (np.diff(eb) >= 0).all() and (np.diff(nb) >= 0).all() and np.diff(eb).max() <= SEG_TILE_ROWS
            real = int(np.searchsorted(nb, nreal, side="left"))           # first boundary that reaches the real node count
            assert nb[real] == nreal and eb[real] == E
            assert (eb[:real + 1] == rp[nb[:real + 1]]).all()
        assert slot.g.meta.seg_tile.shape == ref.meta.seg_tile.shape


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_step_dataset_is_the_step_on_the_collated_batch(kind):
    """Trainer.step_dataset(ds, indices) (collate into the bucket + replay) leaves bitwise the parameters of
    Trainer.step(pad_batch(ds.collate(indices))) — over shuffled epochs that revisit buckets."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer
    if kind == "phonon":
        mk = lambda: _phonon(32, 1)
        cs = synth.phonon_crystals(24, seed=43, dtype=torch.float32)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 1, 200, 41, 2, 32, DEV, 0.0)
        cs = synth.edos_crystals(24, seed=44, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    nmax = max(int(c["x"].shape[0]) for c in cs)
    torch.manual_seed(5)
    m_a = mk().to(DEV)
    m_b = mk()
    m_b.load_state_dict(copy.deepcopy(m_a.state_dict()))
    m_b = m_b.to(DEV)
    ta = Trainer(m_a, lr=1e-3, replay=True, bucket=(64, 1024))
    tb = Trainer(m_b, lr=1e-3, replay=True, bucket=(64, 1024))
    rng = np.random.default_rng(0)
    for epoch in range(3):
        order = rng.permutation(24)
        for i in range(0, 24, 6):
            sel = order[i:i + 6]
            la = ta.step_dataset(ds, sel, n_max=nmax)
            lb = tb.step(collate([cs[j] for j in sel], n_max=nmax).to(DEV))       # host collate: greedy tiles, same bits
            assert float(la) == float(lb), (epoch, i)
    assert ta.slot_hits > 0 and len(ta._slots) < 12
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        assert torch.equal(a, b), k


# ---- attention dropout (VERDICT r1 missing #1; reference: multihead_attention.py:70, --attn_drop utils.py:40) -----------
def _philox_mask_numpy(n, p, seed, stream_id):
    """numpy restatement of dosx_dropout_mask: Philox4x32-10, counter (i/4, stream_id), key = seed, word i%4."""
    nblk = (n + 3) // 4
    b = np.arange(nblk, dtype=np.uint64)
    c = [(b & np.uint64(0xFFFFFFFF)).astype(np.uint64), (b >> np.uint64(32)).astype(np.uint64),
         np.full(nblk, stream_id & 0xFFFFFFFF, np.uint64), np.full(nblk, (stream_id >> 32) & 0xFFFFFFFF, np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    words = np.stack(c, 1).reshape(-1)[:n]
    u = (words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


def test_dropout_mask_kernel():
    from dostransformer_amd import ops
    n, p = 100003, 0.3
    seed = torch.tensor([0x1234567 + (5 << 40)], dtype=torch.int64, device=DEV)
    m = torch.empty(n, device=DEV)
    ops.dropout_mask(m, p, seed, 7)
    ref = _philox_mask_numpy(n, p, int(seed.item()), 7)
    assert np.array_equal(m.cpu().numpy(), ref)
    keep = float((m > 0).float().mean())
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-3          # keep rate
    assert abs(float(m.mean()) - 1.0) < 0.01                                    # unbiased multiplier
    m2 = torch.empty(n, device=DEV)
    ops.dropout_mask(m2, p, seed, 7)
    assert torch.equal(m, m2)                                                   # same (seed, stream) -> same mask
    ops.dropout_mask(m2, p, seed, 8)
    assert not torch.equal(m, m2)                                               # another stream id -> another mask
    seed.add_(1)
    ops.dropout_mask(m2, p, seed, 7)
    assert not torch.equal(m, m2)                                               # bumped seed -> another mask
    z = torch.empty(1000, device=DEV)
    ops.dropout_mask(z, 0.0, seed, 0)
    assert bool((z == 1).all())


@pytest.mark.parametrize("mode", ["cross", "self"])
def test_transformer_encoder_attention_dropout_matches_oracle(mode):
    """TransformerEncoder(attn_dropout=0.3) in training mode: outputs and gradients equal the oracle's with the SAME
    Bernoulli draws (the masks the kernels used, captured per layer); eval mode ignores dropout; p = 0 is untouched."""
    from oracle import dos_oracle as O
    from dostransformer_amd import functional as Fn
    from dostransformer_amd.layers import TransformerEncoder
    torch.manual_seed(0)
    Hh, S, Bq, Nk, T = 32, 51, 5, 9, 2
    enc = TransformerEncoder(embed_dim=Hh, num_heads=1, layers=T, attn_dropout=0.3).to(DEV)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(S, Bq, Hh, generator=gen).to(DEV).requires_grad_(True)
    kv = torch.randn(Nk, Bq, Hh, generator=gen).to(DEV).requires_grad_(True)
    w = torch.randn(S, Bq, Hh, generator=gen).to(DEV)
    Fn.DROP_MASK_LOG = []
    try:
        enc.train()
        y = enc(x, kv, kv) if mode == "cross" else enc(x, x, x)
        masks = [m.clone() for _, _, m in Fn.DROP_MASK_LOG]
    finally:
        Fn.DROP_MASK_LOG = None
    assert len(masks) == T and all(0.5 < float((m > 0).float().mean()) < 0.9 for m in masks)
    (y * w).sum().backward()
    p64 = {"e." + k: v.detach().double().cpu() for k, v in enc.state_dict().items() if v.is_floating_point()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    kv64 = kv.detach().double().cpu().requires_grad_(True)
    m64 = [m.double().cpu() for m in masks]
    yr = O.transformer_encoder(p64, "e", x64, kv64 if mode == "cross" else x64, kv64 if mode == "cross" else x64, T, m64)
    (yr * w.double().cpu()).sum().backward()
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    assert float((x.grad.cpu().double() - x64.grad).abs().max() / x64.grad.abs().max()) < 1e-4
    if mode == "cross":
        assert float((kv.grad.cpu().double() - kv64.grad).abs().max() / kv64.grad.abs().max()) < 1e-4
    # a second training forward draws different masks; eval mode is deterministic and equals the p = 0 oracle
    y2 = enc(x, kv, kv) if mode == "cross" else enc(x, x, x)
    assert not torch.equal(y.detach(), y2.detach())
    enc.eval()
    with torch.no_grad():
        ye = enc(x, kv, kv) if mode == "cross" else enc(x, x, x)
        y0 = O.transformer_encoder(p64, "e", x64, kv64 if mode == "cross" else x64, kv64 if mode == "cross" else x64, T)
    assert float((ye.cpu().double() - y0.detach()).abs().max()) < 2e-5


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


# ---- callable GNN blocks (VERDICT r1 missing #4): fixtures G3 / G4 through the HIP path ---------------------------------
@pytest.mark.parametrize("name", ["mean", "sum"])
def test_g3_processor_block_through_hip(name):
    """One Processor layer called ON ITS OWN with the upstream signature (`DOSTransformer_phonon.py:148-171`): isolated
    node, duplicate edges, unsorted edge_index; outputs, input gradients, parameter gradients, node_mlp_1 untouched."""
    from dostransformer_amd._blocks import EdgeModel, NodeModel, Processor
    from tests.util import load, maxabs, sub
    z = load("g3_processor.npz")
    proc = Processor(EdgeModel(8), NodeModel(8, aggr=name))
    proc.load_state_dict(sub(z, f"{name}/p/"))
    proc = proc.to(DEV)
    x = torch.from_numpy(z["x"]).to(DEV).requires_grad_(True)
    e = torch.from_numpy(z["e"]).to(DEV).requires_grad_(True)
    ei = torch.from_numpy(z["edge_index"]).to(DEV)
    ox, oe = proc(x, ei, e)
    assert maxabs(ox.cpu(), z[f"{name}/ox"]) < 5e-6 and maxabs(oe.cpu(), z[f"{name}/oe"]) < 5e-6
    ((ox * torch.from_numpy(z[f"{name}/wx"]).to(DEV)).sum() + (oe * torch.from_numpy(z[f"{name}/we"]).to(DEV)).sum()).backward()
    assert maxabs(x.grad.cpu(), z[f"{name}/dx"]) < 2e-5 and maxabs(e.grad.cpu(), z[f"{name}/de"]) < 2e-5
    dead = set(str(s) for s in z[f"{name}/dead"])
    for k, v in proc.named_parameters():
        if k in dead:
            assert v.grad is None, k
        else:
            assert maxabs(v.grad.cpu(), z[f"{name}/g/{k}"]) < 5e-5, k
    # the separate EdgeModel / NodeModel calls compose to the same result
    with torch.no_grad():
        e2 = proc.edge_model(x[ei[0]], x[ei[1]], e)
        x2 = proc.node_model(x, ei, e2)
    assert maxabs(e2.cpu(), oe.detach().cpu()) < 2e-6 and maxabs(x2.cpu(), ox.detach().cpu()) < 2e-6


def test_g4_edge_features_through_hip():
    from dostransformer_amd import ops
    from tests.util import load, maxabs
    z = load("g4_edge_features.npz")
    out = ops.edge_feat_sh1(torch.from_numpy(z["edge_vec"]).float().to(DEV), 4.0)
    assert maxabs(out.cpu(), z["edge_attr"]) < 2e-6
    assert out[0].cpu().tolist() == [1.0, 0.0, 0.0, 0.0]                 # zero-length self edge, bit exact


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


def test_reference_style_loop_with_fp64_batches_and_torch_adamw():
    """The reference's own phonon loop (`main_phDOS.py:15-16,101-118`): default dtype float64, `model(batch)` through autograd,
    `MSELoss` against the float64 target, `loss.backward()`, `torch.optim.AdamW.step()` - with the drop-in module.  Outputs
    are fp32 (the kernels' arithmetic), torch promotes them against the fp64 target; three steps track the oracle."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 1, 118, 4, 32, DEV, 0.0)
    p64 = {k: (v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-2)
    crit = torch.nn.MSELoss()
    state = {}
    for step in range(3):
        g64 = synth.phonon_batch(4, seed=700 + step, dtype=torch.float64)
        batch = g64.clone().to(DEV)                                     # float64 fields on the GPU, like upstream
        assert batch.x.dtype == torch.float64
        model.train()
        pg, xn, ps = model(batch)
        loss = torch.sqrt(crit(pg, batch.phdos)).mean() + 1.0 * torch.sqrt(crit(ps, batch.phdos)).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref, _ = O.train_step("phonon", p64, state, g64, 3, 1, lr=1e-4, beta=1.0)
        assert abs(float(loss) - float(ref)) < 5e-5, (step, float(loss), float(ref))
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert float((v.cpu().double() - p64[k]).abs().max()) < 3.1e-4, k
    assert model.alpha.grad is None


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_training_step_is_bitwise_reproducible(kind):
    """No atomics, fixed summation orders, two streams joined by events: the same step on the same inputs gives the same
    bits - gradients and updated parameters - run after run (replay mode, i.e. with the side stream active)."""
    from dostransformer_amd import synth
    from dostransformer_amd.train import Trainer
    if kind == "phonon":
        mk = lambda: _phonon(64, 2)
        g = synth.phonon_batch(16, seed=77, dtype=torch.float32).to(DEV)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 2, 200, 41, 2, 64, DEV, 0.0)
        g = synth.edos_batch(8, seed=78, dtype=torch.float32).to(DEV)
    runs = []
    for _ in range(3):
        torch.manual_seed(4)
        m = mk().to(DEV)
        tr = Trainer(m, lr=1e-3, replay=True)
        for _ in range(3):
            tr.step(g)
        torch.cuda.synchronize()
        runs.append((m.flat_params().grad.clone(), m.flat_params().flat.clone()))
    for gr, fl in runs[1:]:
        assert torch.equal(gr, runs[0][0]) and torch.equal(fl, runs[0][1])


def test_edge_embed_matches_feature_kernel_plus_gemm():
    """dosx_edge_embed_sh1 == dosx_edge_feat_sh1 followed by the K = 4 dosx_gemm (bitwise features, rounding-level z)."""
    from dostransformer_amd import ops
    from tests.util import load
    vec = torch.cat([torch.from_numpy(load("g4_edge_features.npz")["edge_vec"]).float(),
                     (torch.rand(5000, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1) * 2.5]).to(DEV)
    H = 128
    w0 = torch.randn(H, 4, generator=torch.Generator().manual_seed(3)).to(DEV)
    b0 = torch.randn(H, generator=torch.Generator().manual_seed(4)).to(DEV)
    attr, z = ops.edge_embed_sh1(vec, w0, b0, 4.0)
    ref_attr = ops.edge_feat_sh1(vec, 4.0)
    assert torch.equal(attr, ref_attr)
    ref_z = torch.empty(vec.shape[0], H, device=DEV)
    ops.gemm(vec.shape[0], H, [ops.seg(ref_attr)], w0, ref_z, bias=b0)
    assert float((z - ref_z).abs().max()) < 1e-6 * float(ref_z.abs().max())
    z64 = ref_attr.double() @ w0.double().T + b0.double()
    assert float((z.double() - z64).abs().max()) < 2e-6 * float(z64.abs().max())


@pytest.mark.parametrize("H,mean", [(128, True), (64, False), (256, False), (16, True)])
def test_message_gemm_with_segment_sum_epilogue(H, mean):
    """DosxGemm EPI_SEGSUM (second Linear of the edge MLP + scatter_mean / scatter_sum by destination + edge residual in one
    launch, node-aligned row tiles) == the same GEMM followed by dosx_segment_reduce; also on a ghost-padded batch and with
    the residual output switched off (last layer)."""
    from dostransformer_amd import functional as Fn, ops, synth
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    g = synth.phonon_batch(9, seed=5, dtype=torch.float32)
    for padded in (False, True):
        b = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 16, 256)) if padded else g
        m = b.meta.to(DEV)
        N, E = m.num_nodes, m.num_edges
        gen = torch.Generator().manual_seed(1)
        P = {"k.0.weight": torch.randn(2 * H, 3 * H, generator=gen) * 0.1, "k.0.bias": torch.randn(2 * H, generator=gen),
             "k.1.weight": torch.randn(2 * H, generator=gen), "k.1.bias": torch.randn(2 * H, generator=gen),
             "k.2.weight": torch.tensor([0.25]), "k.3.weight": torch.randn(H, 2 * H, generator=gen) * 0.1,
             "k.3.bias": torch.randn(H, generator=gen)}
        P = {k: v.to(DEV) for k, v in P.items()}
        x = torch.randn(N, H, generator=gen).to(DEV)
        e = torch.randn(E, H, generator=gen).to(DEV)
        a = Fn.SegList([ops.seg(x, rmap=ops.rowmap(idx=m.src)), ops.seg(x, rmap=ops.rowmap(idx=m.dst)), ops.seg(e)], [x, e])
        scale = m.inv_deg if mean else None
        msg, _ = Fn.mlp_ln_fwd(P, "k", a, E, H)
        agg0, e0 = torch.empty(N, H, device=DEV), torch.empty(E, H, device=DEV)
        ops.segment_reduce(msg, m.rowptr_dst, scale, agg0, e, e0, N, E, H)
        for with_res in (True, False):
            agg1 = torch.full((N, H), float("nan"), device=DEV)
            e1 = torch.full((E, H), float("nan"), device=DEV) if with_res else None
            out, _ = Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(m.seg_tile, m.rowptr_dst, scale, agg1, e, e1))
            torch.cuda.synchronize()
            assert out is None
            nr = getattr(b, "real_nodes", N)
            assert bool(torch.isfinite(agg1).all())                    # ghost rows included: finite don't-cares
            assert float((agg1[:nr] - agg0[:nr]).abs().max()) <= 2e-6 * float(agg0[:nr].abs().max())
            if with_res:
                assert torch.equal(e1, e0)                             # same fma chain + the same two adds per element


@pytest.mark.parametrize("H,S,B", [(128, 51, 8), (64, 51, 3), (128, 7, 1)])
@pytest.mark.parametrize("mode", ["cross", "self"])
def test_final_layernorm_backward_inside_ffn_bwd(H, S, B, mode, monkeypatch):
    """The encoder's final LayerNorm backward (layers/transformer.py:76-77) fused into the last layer's ffn_bwd launch
    against the stand-alone dosx_layernorm_bwd launch it replaces and against torch autograd on the same module math."""
    from dostransformer_amd import functional as Fn
    from dostransformer_amd.layers import TransformerEncoder
    torch.manual_seed(3)
    enc = TransformerEncoder(embed_dim=H, num_heads=1, layers=2, attn_dropout=0.0).to(DEV)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "layer_norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    gen = torch.Generator().manual_seed(11)
    x0 = torch.randn(S, B, H, generator=gen).to(DEV)
    kv0 = torch.randn(9, B, H, generator=gen).to(DEV)
    w = torch.randn(S, B, H, generator=gen).to(DEV)

    def run(fused):
        monkeypatch.setattr(Fn, "_FUSED_FIN_BWD", fused)
        enc.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        kv = kv0.clone().requires_grad_(True)
        y = enc(x, kv, kv) if mode == "cross" else enc(x, x, x)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        g = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
        return y.detach().clone(), x.grad.clone(), (kv.grad.clone() if mode == "cross" else None), g

    y1, dx1, dkv1, g1 = run(True)
    y0, dx0, dkv0, g0 = run(False)
    sc = lambda t: float(t.abs().max()) + 1e-30
    assert torch.equal(y1, y0)
    assert float((dx1 - dx0).abs().max()) <= 2e-6 * sc(dx0)
    if dkv0 is not None:
        assert float((dkv1 - dkv0).abs().max()) <= 2e-6 * sc(dkv0)
    assert set(g1) == set(g0) and "layer_norm.weight" in g1
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 5e-6 * sc(g0[k]), k


def test_dense_normalize_pool_bwd_is_the_two_launches():
    """dosx_dense_normalize_pool_bwd == dosx_dense_normalize_bwd followed by dosx_graph_pool_bwd, bit for bit (ghost nodes
    included: spare dense row, graph id >= B)."""
    from dostransformer_amd import ops
    torch.manual_seed(0)
    B, H, nmax = 5, 128, 7
    counts = [3, 7, 1, 4, 6]
    N_real = sum(counts)
    N = N_real + 3                                   # 3 ghost (padding) nodes
    node_graph = torch.tensor(sum(([b] * c for b, c in enumerate(counts)), []) + [B] * 3, dtype=torch.int32, device=DEV)
    dense_row = []
    for b, c in enumerate(counts):
        dense_row += [pos * B + b for pos in range(c)]
    dense_row += [nmax * B] * 3
    dense_row = torch.tensor(dense_row, dtype=torch.int32, device=DEV)
    dkv = torch.randn(nmax * B + 1, H, device=DEV)
    kvhat = torch.randn(nmax * B + 1, H, device=DEV)
    rstd = torch.rand(N, device=DEV) + 0.5
    K = 2 * H
    dcat = torch.randn(B, K, device=DEV)
    a = torch.full((N, H), float("nan"), device=DEV)
    ops.dense_normalize_bwd(dkv, kvhat, rstd, dense_row, a, N, H, False, ghost_row=nmax * B)
    ops.graph_pool_bwd(dcat.data_ptr() + 4 * (K - H), K, node_graph, a, N, H, True, num_graphs=B)
    b = torch.full((N, H), float("nan"), device=DEV)
    ops.dense_normalize_pool_bwd(dkv, kvhat, rstd, dense_row, dcat.data_ptr() + 4 * (K - H), K, node_graph, B, b, N, H, False,
                                 ghost_row=nmax * B)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert float(b[N_real:].abs().max()) == 0.0      # ghost nodes: exact zeros


@pytest.mark.parametrize("H,S,B", [(128, 51, 4), (64, 51, 3)])
def test_head_inside_ffn_kernels_matches_standalone_launches(H, S, B, monkeypatch):
    """Final LayerNorm + out_layer in the last ffn_fwd epilogue / first ffn_bwd prologue of the source encoder against the
    ln_rowdot(_bwd) launches they replace: same DOS, same gradients (fp32 rounding of a different summation tree only)."""
    from dostransformer_amd import functional as Fn
    from dostransformer_amd.batch import collate
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(1)
    model = DOSTransformer_phonon(2, 2, 118, 4, H, DEV, 0.0).to(DEV)
    g = collate(synth.phonon_crystals(B, 77, torch.float32)).to(DEV)

    def run(fused):
        monkeypatch.setattr(Fn, "_FUSED_HEAD_FWD", fused)
        monkeypatch.setattr(Fn, "_FUSED_FIN_BWD", fused)
        model.zero_grad(set_to_none=True)
        out = model(g)
        loss = (out[0] ** 2).sum() + 0.5 * (out[2] ** 2).sum()
        loss.backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in (out[0], out[2])], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    o1, g1 = run(True)
    o0, g0 = run(False)
    sc = lambda t: float(t.abs().max()) + 1e-30
    for a, b in zip(o1, o0):
        assert float((a - b).abs().max()) <= 2e-6 * sc(b)
    assert set(g1) == set(g0)
    for k in g0:
        assert float((g1[k] - g0[k]).abs().max()) <= 1e-5 * sc(g0[k]), k
