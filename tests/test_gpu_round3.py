"""Round-3 kernels through the C ABI: finished-mode weight gradients (in-launch reduction over the M-splits by the last
arriving workgroup of every tile, include/dosx.h: DosxWgrad.dst) and the one-launch gradient flush (dosx_grad_flush)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = 2e-5


def ops():
    from dostransformer_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(torch.float32).to(DEV)


def err(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))


def prelu(x, a):
    return torch.where(x >= 0, x, a * x)


def _scratch(o, N, K, ns, bias=True):
    n = o.wgrad_scratch_floats(N, K, ns)
    slab = torch.full((max(n, 1),), float("nan"), device=DEV) if n else None
    slab_b = torch.full((ns * ((N + 63) // 64) * 64,), float("nan"), device=DEV) if (bias and ns > 1) else None
    return slab, slab_b


@pytest.mark.parametrize("M,N,K", [(1000, 256, 384), (333, 128, 118), (70, 64, 64), (5000, 128, 512), (40, 16, 24),
                                    (9344, 256, 384), (6528, 512, 128), (456, 256, 256), (64, 128, 320), (3, 8, 4)])
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


def _mixed_jobs(o):
    jobs = []

    def add(M, N, K, seed, bias=True, **kw):
        dy, a = rnd(M, N, seed=seed), rnd(M, K, seed=seed + 1)
        segs = kw.pop("segs", None) or [o.seg(a)]
        jobs.append(dict(M=M, N=N, K=sum(s.width for s in segs), dy=dy, a=a, segs=segs, kw=kw, bias=bias))

    H = 64
    add(900, 128, 64, 1)
    add(3000, 256, 128, 3, pro=o.PRO_PRELU, pro_alpha=torch.tensor([0.25], device=DEV))
    add(1000, 64, 128, 5, pro=o.PRO_LN_PRELU, pro_gamma=rnd(128, seed=50), pro_beta=rnd(128, seed=51),
        pro_alpha=torch.tensor([0.1], device=DEV))
    add(2000, 256, 64, 7, pro=o.PRO_ROWLN, pro_gamma=rnd(64, seed=52), pro_beta=rnd(64, seed=53),
        pro_stats=torch.rand(2000, 2, device=DEV))
    add(333, 64, 118, 9)                                   # K % 4 != 0: generic staging, scalar dst stores
    x = rnd(50, H, seed=60)
    idx = torch.randint(0, 50, (1200,), device=DEV, dtype=torch.int32)
    e = rnd(1200, H, seed=61)
    add(1200, 128, 2 * H, 11, bias=False, segs=[o.seg(x, rmap=o.rowmap(idx=idx)), o.seg(e)])
    jobs[-1]["keep"] = (x, idx, e)
    for k in range(9):
        add(500 + 100 * k, 64, 64, 20 + 2 * k)
    add(9344, 256, 384, 70)
    add(6528, 128, 512, 72)
    return jobs


def _descs(o, jobs):
    out, descs = [], []
    for j in jobs:
        ns = o.wgrad_splits(j["M"], j["N"], j["K"])
        slab, slab_b = _scratch(o, j["N"], j["K"], ns, j["bias"])
        dw = torch.full((j["N"], j["K"]), float("nan"), device=DEV)
        db = torch.full((j["N"],), float("nan"), device=DEV) if j["bias"] else None
        descs.append(o.wgrad_desc(j["M"], j["N"], o.seg(j["dy"]), j["segs"], slab, slab_b, ns, dst=dw, dst_bias=db, **j["kw"]))
        out.append((dw, db, slab, slab_b))
    return descs, out


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
