"""Data-parallel path on CPU (gloo, world_size 2): the build's sharding + collectives
(dostransformer_amd.dist) reproduce the single-process full-batch loss and gradients when every
rank pads to the global n_max and the phonon loss exchanges its SSE scalars before backward
(SURVEY.md §8e).  The per-rank compute here is the oracle (there is no GPU in this container);
on the GPU the same DataParallel object is driven by dostransformer_amd.train.Trainer."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import dos_oracle as O
        from dostransformer_amd import synth
        from dostransformer_amd.batch import collate
        from dostransformer_amd.dist import DataParallel, shard_batch
        torch.set_num_threads(1)
        torch.manual_seed(0)
        dt = torch.float64
        B, H, L, T = 6, 8, 2, 1
        if kind == "phonon":
            from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
            torch.set_default_dtype(dt)
            model = DOSTransformer_phonon(L, T, 118, 4, H, "cpu", 0.0)
            crystals = synth.phonon_crystals(B, seed=21, dtype=dt)
            fwd = O.dostransformer_phonon_forward
        else:
            from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
            torch.set_default_dtype(dt)
            model = DOSTransformer(L, T, 200, 41, 2, H, "cpu", 0.0)
            crystals = synth.edos_crystals(B, seed=22, dtype=dt)
            fwd = O.dostransformer_forward
        params = {k: v.detach().clone() for k, v in model.state_dict().items() if v.is_floating_point()}
        live = lambda: {k: v.clone().requires_grad_(True) for k, v in params.items()}
        dp = DataParallel()
        g = shard_batch(crystals, world, rank)
        p = live()
        dg, _, ds = fwd(p, g, L, T)
        if kind == "phonon":
            # local SSE -> all-reduce -> every rank scales by the GLOBAL rmse (main_phDOS.py:109-114)
            y = g.phdos
            sse = torch.stack([((dg - y) ** 2).sum(), ((ds - y) ** 2).sum()])
            tot = sse.detach().clone()
            dp.all_reduce_sse(tot)
            count = B * 51                       # static: crystals of the un-sharded batch x bins
            r = torch.sqrt(tot / count)
            # d/dp [ sqrt(SSE_glob/count) ] = dSSE_local / (2 count rmse_glob)
            loss_proxy = (sse[0] / (2 * count * r[0]) + sse[1] / (2 * count * r[1]))
            loss_val = float(r[0] + r[1])
        else:
            nglob = dp.global_count(g.num_graphs)
            assert nglob == B
            yy = torch.where(g.y_ft < 0, torch.zeros_like(g.y_ft), g.y_ft).reshape(g.num_graphs, -1)
            loss_proxy = (torch.sqrt(((yy - dg) ** 2).mean(1)).sum() + torch.sqrt(((yy - ds) ** 2).mean(1)).sum()) / nglob
            lt = loss_proxy.detach().clone().reshape(1)
            td.all_reduce(lt)
            loss_val = float(lt)
        names = [k for k in p]
        grads = torch.autograd.grad(loss_proxy, [p[k] for k in names], allow_unused=True)
        flat = torch.cat([(gr if gr is not None else torch.zeros_like(p[k])).reshape(-1) for k, gr in zip(names, grads)])
        dp.all_reduce_grads(flat)
        if rank == 0:
            # single-process reference on the un-sharded batch
            full = collate(crystals)
            p2 = live()
            dg2, _, ds2 = fwd(p2, full, L, T)
            ref = O.loss_phonon(dg2, ds2, full.phdos) if kind == "phonon" else O.loss_edos(dg2, ds2, full.y_ft)
            g2 = torch.autograd.grad(ref, [p2[k] for k in names], allow_unused=True)
            flat2 = torch.cat([(gr if gr is not None else torch.zeros_like(p2[k])).reshape(-1) for k, gr in zip(names, g2)])
            q.put((loss_val, float(ref), float((flat - flat2).abs().max()), float(flat2.abs().max())))
    finally:
        td.barrier()
        td.destroy_process_group()


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_two_rank_data_parallel_matches_single_process(kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    loss, ref, err, scale = q.get(timeout=10)
    assert abs(loss - ref) < 1e-10
    assert err < 1e-10 * max(1.0, scale)


def test_wrong_nmax_changes_the_result():
    """Sanity of the premise: shards padded to their OWN n_max do not reproduce the full batch."""
    from oracle import dos_oracle as O
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    model = DOSTransformer_phonon(2, 1, 118, 4, 8, "cpu", 0.0)
    p = {k: v.detach() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(5)
    cs = [synth.phonon_crystal(gen, n_atoms=n, dtype=torch.float32) for n in (3, 9)]
    with torch.no_grad():
        full = O.dostransformer_phonon_forward(p, collate(cs), 2, 1)[0]
        own = O.dostransformer_phonon_forward(p, collate(cs[:1]), 2, 1)[0]
        padded = O.dostransformer_phonon_forward(p, collate(cs[:1], n_max=9), 2, 1)[0]
    assert float((own[0] - full[0]).abs().max()) > 1e-4
    assert float((padded[0] - full[0]).abs().max()) < 1e-5


def _minmax_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dostransformer_amd.dist import DataParallel
        dp = DataParallel()
        q.put((rank, dp.min_max(1500), dp.min_max(1500 + rank)))
    finally:
        td.barrier()
        td.destroy_process_group()


def test_min_max_over_ranks_detects_ragged_shards():
    """Trainer.step_dataset's once-per-dataset check: equal per-rank dataset sizes give (n, n), ragged ones do not."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_minmax_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert [g[1] for g in got] == [(1500, 1500)] * 2 and [g[2] for g in got] == [(1500, 1501)] * 2


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dostransformer_amd._fused import FlatParams
        from dostransformer_amd.dist import DataParallel
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        torch.manual_seed(0)
        fp = FlatParams(DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0), torch.device("cpu"))
        dp = DataParallel()
        gen = torch.Generator().manual_seed(100 + rank)
        fp.grad.copy_(torch.randn(fp.total, generator=gen))
        whole = fp.grad.clone()
        dp.all_reduce_grads(whole)
        # the order train.Trainer issues them in: early (backward reaches the GNN), mid (backward reaches layer 0), last (end)
        early = dp.all_reduce_grads_async(fp.grad[fp.n_late:])
        mid = dp.all_reduce_grads_async(fp.grad[fp.n_last:fp.n_late])
        dp.all_reduce_grads(fp.grad[:fp.n_last])
        mid.wait()
        early.wait()
        q.put((rank, bool(torch.equal(fp.grad, whole)), fp.n_last, fp.n_late, fp.total))
    finally:
        td.barrier()
        td.destroy_process_group()


def test_three_gradient_buckets_cover_the_flat_buffer_exactly_once():
    """the flat gradient is all-reduced as [last | mid | early] slices (two of them asynchronously, started
    during the backward pass); together they are the all-reduce of the whole buffer, on 2 gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert all(g[1] for g in got) and all(0 < g[2] < g[3] < g[4] for g in got)
