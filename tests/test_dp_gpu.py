"""Data-parallel `Trainer` with MORE THAN ONE RANK on real hardware.

The pool's boxes have one MI355X, so the two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one
device); `dist.DataParallel` stages the sums through the host in that case.  What runs with world_size 2 is the
production code: `shard_batch` (global n_max, recorded global batch size), the phonon SSE pre-reduce before backward
(`main_phDOS.py:109-114`: ONE rmse over all B*51 elements), the early-bucket hook under the GNN backward, the split
recording of a replayed step around the collectives, and the two-bucket `optimizer_step`.  Checked against a
single-process `Trainer` on the un-sharded batches."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(out, world, suite):
    port = _free_port()
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    logf = [open(os.path.join(out, f"rank{r}.log"), "wb") for r in range(world)]     # (files, not pipes: 8 chatty ranks must not block)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), str(world), str(port), str(out), suite],
                              env=env, cwd=ROOT, stdout=logf[r], stderr=subprocess.STDOUT) for r in range(world)]
    try:
        for p in procs:
            p.wait(timeout=1500)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        raise
    finally:
        for f in logf:
            f.close()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{open(os.path.join(out, f'rank{r}.log'), errors='replace').read()[-4000:]}"
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(world)]


@pytest.fixture(scope="module")
def two_rank_run(tmp_path_factory):
    return _run_ranks(str(tmp_path_factory.mktemp("dp2")), 2, "small")


@pytest.fixture(scope="module")
def eight_rank_full_run(tmp_path_factory):
    return _run_ranks(str(tmp_path_factory.mktemp("dp8")), 8, "full")


@pytest.mark.parametrize("kind,mode", [("phonon", "eager"), ("phonon", "replay"), ("edos", "eager"), ("edos", "replay"),
                                       ("phonon", "replay_mid"), ("edos", "replay_mid")])
def test_two_rank_trainer_matches_single_process(two_rank_run, kind, mode):
    from dostransformer_amd.batch import collate
    from dostransformer_amd.train import Trainer
    from tests.dp_worker import B_GLOBAL, make_crystals, make_model
    r0, r1 = two_rank_run
    pre = f"{kind}/{mode}/"
    # replicas stay in lockstep: identical parameters and all-reduced gradients on both ranks
    for k in r0.files:
        if k.startswith(pre) and not k.endswith("/loss"):
            assert np.array_equal(r0[k], r1[k]), k
    dev = "cuda:0"
    model = make_model(kind, dev)
    tr = Trainer(model, lr=1e-3, beta=1.0)
    losses, grad0 = [], None
    for step in (0, 1, 0):
        g = collate(make_crystals(kind, step)).to(dev)
        losses.append(float(tr.step(g)))
        if grad0 is None:
            torch.cuda.synchronize()
            grad0 = model.flat_params().grad.detach().cpu().numpy().copy()
    # the phonon loss is global (every rank reports it); the eDOS loss is a mean over crystals: ranks report their share
    dp_loss = r0[pre + "loss"] if kind == "phonon" else r0[pre + "loss"] + r1[pre + "loss"]
    assert np.allclose(dp_loss, losses, rtol=2e-5, atol=2e-6), (dp_loss, losses)
    # gradients of step 0 (sum over ranks of shard gradients scaled by the GLOBAL count) == full-batch gradients
    gd = r0[pre + "grad0"]
    assert gd.shape == grad0.shape
    assert np.abs(gd - grad0).max() <= 2e-5 * np.abs(grad0).max(), np.abs(gd - grad0).max() / np.abs(grad0).max()
    # parameters after 3 AdamW steps
    fp = model.flat_params()
    for k, v in model.state_dict().items():
        if not v.is_floating_point():
            continue
        a, b = r0[pre + "p/" + k], v.detach().cpu().numpy()
        if k in fp.G:
            # Adam's update is ~lr*sign(g) where |g| is at the fp32 noise floor of the two summation orders (the gradient
            # check above is the sharp one: 2e-5 of the tensor maximum): compare the parameters where the gradient is
            # resolved - 3 steps of lr 1e-3 move an element by up to 3e-3, 2e-5 is < 1 % of that - and bound the rest
            # by the step size
            gk = fp.G[k].detach().cpu().numpy()
            ok = np.abs(gk) >= 1e-2 * np.abs(gk).max()
            assert np.abs(a - b)[ok].max() <= 2e-5, (k, np.abs(a - b)[ok].max())
            assert np.abs(a - b).max() <= 3.1e-3, k
        else:
            assert np.array_equal(a, b), k          # dead parameters: untouched everywhere


def test_bench_two_ranks_share_gpu(tmp_path):
    """bench.py's N > 1 path (sharding, barrier, max-over-ranks timing, rank-0 JSON line) executed with 2 ranks on the
    one GPU of the box (gloo; no scaling claim — this only proves the code path runs and reports the whole-job rate)."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                                       "--config", "phonon_h64_b8", "--no-cpu-baseline", "--dist-backend", "gloo", "--share-gpu"],
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        outs.append((p.returncode, o.decode(), e.decode(errors="replace")))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f"rank {r}: {e[-3000:]}"
    rec = json.loads(outs[0][1].strip().splitlines()[-1])
    assert outs[1][1].strip() == ""                       # only rank 0 prints
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and abs(rec["value"] - 16 * rec["steps"] / (rec["ms_per_step"] * 1e-3 * rec["steps"])) < 1e-2 * rec["value"]


@pytest.mark.parametrize("kind,mode", [("phonon", "replay"), ("edos", "replay")])
def test_eight_rank_full_size_configs_match_single_process(eight_rank_full_run, kind, mode):
    """BASELINE.json configs[3] (Phonon-DOS L3 T2 H128, global batch 512) and configs[4] (Electron-DOS H256 T4, global batch
    256) at FULL global size with the 8 ranks of the 8-GPU configuration - 8 fresh processes that share the box's one
    MI355X over gloo, each training on its `shard_batch` shard (64 resp. 32 crystals) - against a single-process `Trainer` on
    the un-sharded batch (it fits one GPU): global loss (`main_phDOS.py:109-114` is ONE rmse over all B*51 elements), the
    all-reduced step-0 gradients, the parameters after 2 AdamW steps (step 0 runs eagerly while the replay plan - split
    around the SSE and early-bucket collectives - is recorded, step 1 replays it); replicas bit-identical to each other.  No scaling claim
    (one GPU, host-staged sums): this proves the 8-way sharded step computes the full-batch step."""
    from dostransformer_amd.batch import collate
    from dostransformer_amd.train import Trainer
    from tests.dp_worker import SUITES, make_crystals, make_model, suite_steps
    ranks = eight_rank_full_run
    pre = f"{kind}/{mode}/"
    for r in ranks[1:]:
        for k in ranks[0].files:
            if k.startswith(pre) and not k.endswith("/loss"):
                assert np.array_equal(ranks[0][k], r[k]), k
    dev = "cuda:0"
    from dostransformer_amd import functional as Fn
    model = make_model(kind, dev, "full")
    H = model._cfg.H
    assert H == SUITES["full"][kind][2]
    # The EdgeModel's first Linear is evaluated in its factored form from a size limit on (functional._factor_edge): the same
    # sums in another order.  The reference run takes the form the RANKS take on their shards, so that the strict tolerances
    # below keep measuring the sharding alone; the full batch's own form (if it differs) is compared right after, at the
    # tolerance of the fp32-vs-fp64 oracle comparison (tests/test_gpu_models.py: GRAD_TOL).
    g0 = collate(make_crystals(kind, SUITES["full"]["steps"][0], "full"))
    policy = (Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_LAST, Fn._FACTOR_LAST_MIN_GF, Fn._FACTOR_HEADS, Fn._FACTOR_HEADS_MIN_GF)
    S_, B1 = model._cfg.S, g0.meta.num_graphs
    E8, E1 = g0.meta.num_edges // 8, g0.meta.num_edges
    shard_form = (Fn._factor_edge(E8, H), Fn._factor_last(E8, H), Fn._factor_heads(S_ * B1 // 8, H))
    full_form = (Fn._factor_edge(E1, H), Fn._factor_last(E1, H), Fn._factor_heads(S_ * B1, H))
    try:
        Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_LAST, Fn._FACTOR_LAST_MIN_GF = shard_form[0], 0.0, shard_form[1], 0.0
        Fn._FACTOR_HEADS, Fn._FACTOR_HEADS_MIN_GF = shard_form[2], 0.0
        tr = Trainer(model, lr=1e-3, beta=1.0)
        losses, grad0 = [], None
        for step in suite_steps("full", kind):
            g = collate(make_crystals(kind, step, "full")).to(dev)
            assert g.num_graphs == SUITES["full"][kind][3]
            losses.append(float(tr.step(g)))
            if grad0 is None:
                torch.cuda.synchronize()
                grad0 = model.flat_params().grad.detach().cpu().numpy().copy()
    finally:
        Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_LAST, Fn._FACTOR_LAST_MIN_GF, Fn._FACTOR_HEADS, Fn._FACTOR_HEADS_MIN_GF = policy
    if full_form != shard_form:
        m2 = make_model(kind, dev, "full")
        Trainer(m2, lr=1e-3, beta=1.0).forward_backward(g0.to(dev))
        torch.cuda.synchronize()
        fp2 = m2.flat_params()
        g2 = fp2.grad.detach().cpu().numpy()
        assert np.abs(g2 - grad0).max() <= 3e-4 * np.abs(grad0).max()
        for k, o, t in zip(fp2.names, fp2.offsets, [fp2.G.get(n) for n in fp2.names]):
            if t is not None:
                a_, b_ = g2[o:o + t.numel()], grad0[o:o + t.numel()]
                assert np.abs(a_ - b_).max() <= 3e-3 * (np.abs(b_).max() + 1e-12), ("full batch, default form", k)
    dp_loss = ranks[0][pre + "loss"] if kind == "phonon" else sum(r[pre + "loss"] for r in ranks)
    assert np.allclose(dp_loss, losses, rtol=5e-5, atol=5e-6), (dp_loss, losses)
    gd = ranks[0][pre + "grad0"]
    assert gd.shape == grad0.shape
    rel = np.abs(gd - grad0).max() / np.abs(grad0).max()
    assert rel <= 1e-4, rel
    # typical error, not just the maximum: the 99th percentile of the element errors stays two orders below the maximum bound
    assert np.percentile(np.abs(gd - grad0), 99) <= 1e-5 * np.abs(grad0).max()
    fp = model.flat_params()
    off = dict(zip(fp.names, fp.offsets))
    bad = []
    worst_t = worst_p = (0.0, None)
    for k, v in model.state_dict().items():
        if not v.is_floating_point():
            continue
        a, b = ranks[0][pre + "p/" + k], v.detach().cpu().numpy()
        if k in fp.G:
            n = fp.G[k].numel()
            g0s, g0d = grad0[off[k]:off[k] + n].reshape(b.shape), gd[off[k]:off[k] + n].reshape(b.shape)
            # per TENSOR: the 8-way sharded, all-reduced gradient against the full-batch one, relative to the tensor's maximum
            rel_t = float(np.abs(g0d - g0s).max() / (np.abs(g0s).max() + 1e-12))
            worst_t = max(worst_t, (rel_t, k))
            if n >= 1000:           # ... and its TYPICAL element: 99 % of a tensor's elements agree to 1e-3 of its maximum
                worst_p = max(worst_p, (float(np.percentile(np.abs(g0d - g0s), 99) / (np.abs(g0s).max() + 1e-12)), k))
            # Adam moves an element by ~lr*sign(g) per step, so parameters are compared where the gradient is resolved in
            # BOTH steps (an element whose step-0 gradient is at the noise floor of the two summation orders may legitimately
            # go the other way in step 0) and bounded by the two steps' movement elsewhere
            gk = fp.G[k].detach().cpu().numpy()
            ok = (np.abs(gk) >= 5e-2 * np.abs(gk).max()) & (np.abs(g0s) >= 5e-2 * np.abs(g0s).max())
            if ok.any() and np.abs(a - b)[ok].max() > 2e-4:              # 10 % of the two steps' movement
                bad.append((k, "resolved", float(np.abs(a - b)[ok].max())))
            # an element moves by at most ~lr per step, so two runs are at most 2 steps x 2 lr apart (an element at the noise
            # floor of both gradients may go opposite ways in both steps)
            if np.abs(a - b).max() > 4.2e-3:
                bad.append((k, "bound", float(np.abs(a - b).max())))
        elif not np.array_equal(a, b):                                    # dead parameters: untouched everywhere
            bad.append((k, "dead", 0.0))
    print(f"8-rank {kind}: worst per-tensor gradient error {worst_t[0]:.3e} at {worst_t[1]}, worst 99th percentile {worst_p[0]:.3e} at {worst_p[1]}")
    # The ranks' 64-crystal shards and the 512-crystal reference run take different kernel forms for the same layer (attention
    # inside / outside the feed-forward launch, tile heights: functional.encoder_fwd's size policies), i.e. other fp32 summation
    # orders - and a pre-activation within rounding of zero then gates its ReLU differently in the two runs: single ELEMENTS of a
    # weight gradient differ by a finite amount (DESIGN.md 6 / HISTORY.md 4: the gate-flip analysis; 1.3e-3 .. 1.4e-3 of the tensor's
    # maximum observed in rounds 4 and 6).  So the MAXIMUM is held to the tolerance of the fp32-vs-fp64 oracle comparison
    # (tests/test_gpu_models.py: GRAD_TOL), 99 % of a tensor's elements to GRAD_TOL_P99 - a wrong shard weight or a missing rank
    # moves every element by ~1/8 of its value
    assert worst_t[0] <= 3e-3, worst_t
    assert worst_p[0] <= 1e-3, worst_p          # (tests/test_gpu_models.py: GRAD_TOL_P99; observed 2.6e-4 at embeddings.weight)
    assert not bad, bad


class _DelayedComm:
    """Stand-in for the RCCL collective of a 2-rank job on a 1-GPU box, with NCCL's stream semantics: the collective runs on
    a stream of the communicator, ordered after the work already queued on the caller's stream; `async_op=True` returns a
    handle whose `wait()` makes the then-current stream wait for it.  STRICTER than NCCL in one respect: every collective
    gets a stream of its own, so a later (synchronous) collective does not order the caller behind an earlier asynchronous
    one by accident - only the handle's wait() does.  The "sum over ranks" adds a second rank whose contribution is the
    constant 0.01, `delay` GPU cycles late."""

    def __init__(self, real, delay):
        self.real, self.delay, self.streams, self.calls, self.waits = real, int(delay), [], 0, 0

    def __call__(self, t, op=None, group=None, async_op=False):
        if not t.is_cuda:
            return self.real(t, op=op, group=group, async_op=async_op)
        self.calls += 1
        comm = torch.cuda.Stream()
        self.streams.append(comm)
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            if self.delay:
                torch.cuda._sleep(self.delay)
            t.add_(0.01)               # (not a scaling: Adam's update is invariant to scaling every gradient)

        outer = self

        class Work:
            def wait(self_w):
                outer.waits += 1
                torch.cuda.current_stream().wait_stream(comm)
                return True
        w = Work()
        if async_op:
            return w
        w.wait()
        return None


def test_optimizer_waits_for_a_late_early_bucket_allreduce_under_replay(monkeypatch):
    """The NCCL branch of `DataParallel` (device buffers, `all_reduce(async_op=True)` for the early gradient bucket, started
    on the weight-gradient stream underneath the GNN backward, `train.Trainer._mid_hook`) under REPLAY, with the collective
    finishing ~40 ms late on the communicator's stream: `optimizer_step` must wait for it.  World size 1 (one GPU), the
    collective replaced by a delayed "+ 0.01" with NCCL's stream semantics.  Same parameters, bit for bit, with and without the
    delay, and every handle's wait() was called (counted by the stand-in)."""
    import torch.distributed as td
    from dostransformer_amd import dist as D
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    created = not td.is_initialized()
    if created:
        td.init_process_group("nccl", rank=0, world_size=1)
    dev = "cuda:0"
    try:
        g = synth.phonon_batch(6, seed=21, dtype=torch.float32)
        g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges)).to(dev)
        real = td.all_reduce

        def run(delay, steps):
            comm = _DelayedComm(real, delay)
            monkeypatch.setattr(D.td, "all_reduce", comm)
            torch.manual_seed(0)
            model = DOSTransformer_phonon(3, 1, 118, 4, 32, dev, 0.0).to(dev)
            dp = D.DataParallel()
            assert not dp.staged                                   # the NCCL branch
            tr = Trainer(model, lr=1e-3, dist=dp, replay=True)
            for _ in range(steps):
                tr.step(g, 12)                                     # "two ranks" of 6 crystals
            torch.cuda.synchronize()
            monkeypatch.setattr(D.td, "all_reduce", real)
            assert comm.calls >= 3 * steps                         # SSE pair, early bucket, GNN bucket per step
            assert comm.waits >= 3 * steps                         # ... and every one of them was waited for
            fp = model.flat_params()
            assert 0 < fp.n_last < fp.n_late < fp.total and tr._early_work is None and tr._mid_work is None
            return {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if v.is_floating_point()}

        ref = run(0, 4)                        # step 0 records the plan (split around the collectives), steps 1-3 replay it
        late = run(int(1e8), 4)
        for k in ref:
            assert torch.equal(ref[k], late[k]), k
        # (A negative control by VALUE - the handle's wait() disabled - is not observable on this stack: HIP multiplexes
        #  streams onto a few hardware queues, the communicator's stream can share the caller's queue and then orders it
        #  anyway.  The count above is the control: every collective's handle was waited for, under replay too.)
    finally:
        if created:
            td.destroy_process_group()
