"""The training step's host plumbing on the GPU: shape buckets, the recorded launch lists and what feeds them."""
import copy

import pytest
import torch

import ctypes as C
import math
import numpy as np
import torch.nn.functional as F
from tests.gpu_util import (DEV, TOL, _FakeDist, _Hog, _attn_ref, _descs, _fat_crystals, _fatten, _graph, _mixed_jobs, _node_block, _philox_mask_numpy, _phonon, _random_crystals, _reduce, _ref, _scratch, _sliver_case, err, ops, prelu, rnd)  # noqa: F401
pytestmark = pytest.mark.gpu
DEV = "cuda"


def _phonon(H=32, T=1):
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(3)
    return DOSTransformer_phonon(3, T, 118, 4, H, DEV, 0.0).to(DEV)


def test_revisited_batch_is_not_copied_again_but_an_edited_one_is():
    """Round 6: a bucket remembers which batch object its static buffers hold (identity + torch's in-place version counters of
    every field); stepping on the same, untouched batch again issues no copy launch, a batch whose field was written in place -
    or another batch of the same bucket - is copied as before.  Same parameters as a trainer that always copies."""
    from dostransformer_amd import ops, synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    from dostransformer_amd.train import Trainer, _Slot
    cs = synth.phonon_crystals(6, seed=41, dtype=torch.float32)
    g = collate(cs)
    g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 8, 128)).to(DEV)
    g2 = copy.copy(g)                                   # another batch OBJECT of the same bucket (same tensors: same numbers)
    m_a, m_b = _phonon(), _phonon()
    m_b.load_state_dict(copy.deepcopy(m_a.state_dict()))
    ta, tb = Trainer(m_a, lr=1e-3, replay=True), Trainer(m_b, lr=1e-3, replay=True)
    calls = []
    real = ops.copy_many
    always = _Slot.load

    def counting(pairs):
        calls.append(len(pairs))
        return real(pairs)

    def load_always(self, gg):                          # the reference trainer: forget what the bucket holds
        self._loaded = None
        return always(self, gg)
    ops.copy_many = counting
    try:
        for step in range(6):
            if step == 3:
                g.phdos.mul_(0.5)                       # in-place edit of a field: the next step must see it
            batch = g2 if step == 5 else g
            n0 = len(calls)
            la = float(ta.step(batch))
            n_copy = len(calls) - n0
            _Slot.load = load_always
            try:
                lb = float(tb.step(batch))
            finally:
                _Slot.load = always
            assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb)), (step, la, lb)
            # step 0 records (the slot clones the batch); 1, 2, 4: untouched revisit -> no copy; 3: edited -> copy; 5: other object -> copy
            assert n_copy == (1 if step in (3, 5) else 0), (step, n_copy)
    finally:
        ops.copy_many = real
    torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        assert torch.equal(a, b), k


def test_fresh_batches_at_recycled_addresses_are_copied():
    """A training loop that collates a NEW batch object every step and drops the previous one: CPython hands the new object the old
    `id()` and the caching allocator the old device pointers (version counters 0 again), so an address-based "the bucket already
    holds this batch" check skips the copy and the step silently trains on stale data (found by
    test_step_dataset_is_the_step_on_the_collated_batch in round 6).  The bucket compares the objects themselves."""
    from dostransformer_amd import ops, synth
    from dostransformer_amd.train import Trainer, _Slot
    n_atoms = [3, 5, 2, 7, 4, 6]
    m_a, m_b = _phonon(), _phonon()
    m_b.load_state_dict(copy.deepcopy(m_a.state_dict()))
    ta, tb = Trainer(m_a, lr=1e-3, replay=True), Trainer(m_b, lr=1e-3, replay=True)
    always = _Slot.load

    def load_always(self, gg):
        self._loaded = None
        return always(self, gg)
    ptrs = set()
    for k in range(8):
        g = synth.phonon_batch(6, seed=90 + k, dtype=torch.float32, n_atoms=n_atoms).to(DEV)
        ptrs.add(g.x.data_ptr())
        la = float(ta.step(g))
        _Slot.load = load_always
        try:
            lb = float(tb.step(g))
        finally:
            _Slot.load = always
        assert la == lb, (k, la, lb)
        del g
    assert len(ta._slots) == 1                          # one bucket throughout: every step after the first is a load + replay
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        assert torch.equal(a, b), k


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


def test_step_dataset_checks_shard_sizes_once_per_dataset():
    """without an explicit n_global, step_dataset assumes B * world crystals in the un-sharded batch -
    true for every batch of an epoch iff all ranks hold equally many crystals.  That is verified once per dataset with one
    min/max over the ranks; ragged shards are refused with a message that asks for n_global; an explicit n_global is taken
    as is."""
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    ds = DeviceDataset(synth.phonon_crystals(16, 3, torch.float32), DEV)
    nmax = 12
    model = DOSTransformer_phonon(2, 1, 118, 4, 64, DEV, 0.0).to(DEV)
    d = _FakeDist((16, 16))
    tr = Trainer(model, replay=True, dist=d)
    for _ in range(3):
        tr.step_dataset(ds, list(range(8)), n_max=nmax)
    torch.cuda.synchronize()
    assert d.calls == 1                                        # once per dataset, not per step
    assert all(k[4] == 16 for k in tr._slots)                  # n_global = B * world in the bucket key
    ragged = _FakeDist((16, 17))
    tr2 = Trainer(DOSTransformer_phonon(2, 1, 118, 4, 64, DEV, 0.0).to(DEV), replay=True, dist=ragged)
    with pytest.raises(ValueError, match="n_global"):
        tr2.step_dataset(ds, list(range(8)), n_max=nmax)
    tr2.step_dataset(ds, list(range(8)), n_global=15, n_max=nmax)      # the caller knows: no check, its count is used
    torch.cuda.synchronize()
    assert ragged.calls == 1 and all(k[4] == 15 for k in tr2._slots)


def test_checkpointed_dropout_seed_is_rank_independent(monkeypatch):
    """Trainer.state_dict stores the dropout seed WITHOUT the saving rank's offset; a rank that loads it
    re-applies its own offset, so resumed data-parallel ranks keep drawing different masks for their different shards - the
    masks their uninterrupted selves would have drawn."""
    from dostransformer_amd import _models, synth, train
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    off = lambda r: (0x9E3779B97F4A7C15 * (r + 1)) % _models._SEED_MOD
    g = collate(synth.phonon_crystals(4, 1, torch.float32)).to(DEV)
    seeds = {}
    for r in (0, 3):
        monkeypatch.setattr(_models, "rank_seed_offset", lambda r=r: off(r))
        monkeypatch.setattr(train, "rank_seed_offset", lambda r=r: off(r))
        torch.manual_seed(11)
        model = DOSTransformer_phonon(2, 1, 118, 4, 64, DEV, 0.25).to(DEV)
        tr = train.Trainer(model)
        model.train()
        for _ in range(2):
            tr.step(g)
        seeds[r] = (int(model._drop_seed.item()), tr.state_dict())
    assert seeds[0][0] != seeds[3][0] and seeds[0][1]["drop_seed_base"] == seeds[3][1]["drop_seed_base"]
    # rank 3 resumes from the file rank 0 wrote: it gets ITS seed back, not rank 0's
    monkeypatch.setattr(train, "rank_seed_offset", lambda: off(3))
    monkeypatch.setattr(_models, "rank_seed_offset", lambda: off(3))
    torch.manual_seed(99)
    model = DOSTransformer_phonon(2, 1, 118, 4, 64, DEV, 0.25).to(DEV)
    tr = train.Trainer(model)
    tr.load_state_dict(seeds[0][1])
    assert int(model._drop_seed.item()) == seeds[3][0]


def test_counter_pool_never_aliases_launches_in_flight(monkeypatch):
    """recorded programs own their arrival counters; the eager ring synchronises before it hands an entry
    out a second time and grows for a request larger than itself; a failed call drops the ring."""
    o = ops()
    pool = o._CounterPool()
    monkeypatch.setattr(pool, "SIZE", 64)
    syncs = []
    real_sync = torch.cuda.synchronize
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: (syncs.append(1), real_sync(*a, **k))[1])
    a = pool.take(DEV, 40)
    b = pool.take(DEV, 20)
    assert b == a + 160 and not syncs
    c = pool.take(DEV, 10)                     # 40 + 20 + 10 > 64: wraps - after a device synchronisation
    assert c == a and len(syncs) == 1
    big = pool.take(DEV, 1000)                 # larger than the ring: a new, larger ring (zeroed), not an error
    assert len(syncs) == 2 and pool._bufs[str(DEV)][0].numel() >= 1000 and big == pool._bufs[str(DEV)][0].data_ptr()
    assert int(pool._bufs[str(DEV)][0].abs().sum()) == 0
    o.RECORDER.begin()
    try:
        r1, r2 = pool.take(DEV, 8), pool.take(DEV, 8)
        own = list(o.RECORDER.keep)
    finally:
        o.RECORDER.end()
    ring = pool._bufs[str(DEV)][0]
    lo, hi = ring.data_ptr(), ring.data_ptr() + 4 * ring.numel()
    assert r1 != r2 and not (lo <= r1 < hi) and not (lo <= r2 < hi) and len(own) == 2
    pool.poison()
    assert pool._bufs == {}


def test_edos_example_driver_end_to_end(tmp_path):
    """examples/train_edos.py (counterpart of `main_eDOS.py:101-175` with the flags of `utils.py:25-43`):
    DeviceDataset -> Trainer.step_dataset (replay) -> evaluate.test(Predictor) at batch size 1 every --eval epochs -> test
    split on a new best -> checkpoint.  The loss must go down, the metrics must be finite, and the checkpoint must reload
    into a fresh module and reproduce the saved model's predictions."""
    import importlib.util
    import os
    from dostransformer_amd import checkpoint, synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_edos", os.path.join(root, "examples", "train_edos.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "best.pt")
    res = mod.main(["--epochs", "6", "--eval", "2", "--crystals", "100", "--hidden", "32", "--transformer", "1",
                    "--batch_size", "16", "--lr", "2e-3", "--out", out])
    h = res["train_loss"]
    assert len(h) == 6 and all(np.isfinite(h)) and h[-1] < 0.9 * h[0], h
    assert res["best_epoch"] in (2, 4, 6) and np.isfinite(res["best_valid_rmse"]) and all(np.isfinite(res["test"]))
    assert os.path.exists(out)
    fresh = DOSTransformer(3, 1, 200, 41, 2, 32, DEV, 0.0).to(DEV)
    extra = checkpoint.load(out, fresh)
    assert extra["epoch"] == res["best_epoch"]
    g = collate(synth.edos_crystals(3, 5, torch.float32)).to(DEV)
    fresh.eval()
    with torch.no_grad():
        a = fresh(g)[2]
    assert bool(torch.isfinite(a).all())


def test_first_time_bucket_runs_in_a_live_one():
    """Trainer(promote=...) + step_dataset: a shape bucket asked for the first time borrows the smallest live bucket that holds
    it (ghost rows are exact don't-cares whatever the padding) instead of recording a launch list of its own; the second time it
    is asked for, it records.  Losses and parameters against a trainer without promotion (every bucket its own program): the same
    step up to the summation order of the weight-gradient splits (padding moves the split points)."""
    import copy
    from dostransformer_amd import synth
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import Trainer
    cs = synth.phonon_crystals(40, seed=91, dtype=torch.float32)
    ds = DeviceDataset(cs, DEV)
    nmax = max(int(c["x"].shape[0]) for c in cs)
    order = sorted(range(40), key=lambda i: int(cs[i]["x"].shape[0]))
    big, small = order[-8:], order[:8]                       # 8 largest / 8 smallest crystals: different (N, E) buckets
    torch.manual_seed(2)
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    _phonon = lambda H, T: DOSTransformer_phonon(3, T, 118, 4, H, DEV, 0.0)
    m_a = _phonon(32, 1).to(DEV)
    m_b = _phonon(32, 1)
    m_b.load_state_dict(copy.deepcopy(m_a.state_dict()))
    m_b = m_b.to(DEV)
    ta = Trainer(m_a, lr=1e-3, replay=True, bucket=(8, 64), promote=10.0)        # (any live bucket that is large enough)
    tb = Trainer(m_b, lr=1e-3, replay=True, bucket=(8, 64))
    seq = [big, small, big, small, small]
    for k, sel in enumerate(seq):
        la, lb = ta.step_dataset(ds, sel, n_max=nmax), tb.step_dataset(ds, sel, n_max=nmax)
        assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb))), k
    # small: first sighting -> ran in big's bucket (promoted), second sighting -> recorded its own, third -> exact hit
    assert ta.slot_promoted == 1 and len(ta._slots) == 2 and ta.slot_misses == 2
    assert tb.slot_promoted == 0 and len(tb._slots) == 2 and tb.slot_misses == 2
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        if a.is_floating_point():
            assert float((a - b).abs().max()) < 5e-3, k          # 5 steps of lr 1e-3 bound any element's drift
            assert float((a - b).abs().median()) < 1e-5, k


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
