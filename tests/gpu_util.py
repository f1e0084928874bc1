"""Helpers shared by the GPU test files (tests/test_gpu_*.py): seeded inputs, error measures, small reference forms."""
import copy
import ctypes as C
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from tests.util import rmse  # noqa: F401

DEV = "cuda"
TOL = 2e-5


def ops():
    from dostransformer_amd import ops as o
    return o

def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).to(torch.float32).to(DEV)

def err(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))

def prelu(x, a):
    return torch.where(x >= 0, x, a * x)

def _reduce(o, sink):
    sink.flush()
    torch.cuda.synchronize()

def _attn_ref(x, kvhat, gam, bet, Sq, Bq, Nk, Bk, H, qs, qb, mask=None):
    """float64 torch reference of the attention block (same math as oracle.encoder_layer's first half); mask: the
    attention-dropout multiplier applied to the softmax output (multihead_attention.py:70)."""
    rows = (torch.arange(Sq, device=DEV)[:, None] * qs + torch.arange(Bq, device=DEV)[None, :] * qb).reshape(-1)
    xq = x[rows].reshape(Sq, Bq, H)
    q = F.layer_norm(xq, (H,), gam, bet, 1e-5)
    k = (kvhat * gam + bet).reshape(Nk, Bk, H)
    k = k[:, torch.arange(Bq, device=DEV) % Bk]
    w = torch.bmm(q.transpose(0, 1), k.permute(1, 2, 0)) * H ** -0.5
    p = torch.softmax(w, -1)
    pd = p if mask is None else p * mask
    out = xq + torch.bmm(pd, k.transpose(0, 1)).transpose(0, 1)
    return out.reshape(Sq * Bq, H), p

# ---- §8f-3 periodic neighbour list (dosx_neighbor_count / _fill) --------------------------------------------------
def _random_crystals(seed, sizes):
    rng = np.random.default_rng(seed)
    pos, cells = [], []
    for n in sizes:
        cell = np.diag(rng.uniform(2.5, 6.0, 3)) + rng.uniform(-1.2, 1.2, (3, 3))       # triclinic, well conditioned
        frac = rng.uniform(-1.5, 2.5, (n, 3))                                           # NOT wrapped into the cell
        pos.append(frac @ cell)
        cells.append(cell)
    return pos, cells

def _phonon(H=64, T=2):
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    return DOSTransformer_phonon(3, T, 118, 4, H, DEV, 0.0)

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

def _scratch(o, N, K, ns, bias=True):
    n = o.wgrad_scratch_floats(N, K, ns)
    slab = torch.full((max(n, 1),), float("nan"), device=DEV) if n else None
    slab_b = torch.full((ns * ((N + 63) // 64) * 64,), float("nan"), device=DEV) if (bias and ns > 1) else None
    return slab, slab_b

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
    # long jobs: 128 x 64 tiles, every prologue and a gathered operand
    add(17880, 256, 128, 80, pro=o.PRO_PRELU, pro_alpha=torch.tensor([0.25], device=DEV))
    add(16384, 128, 256, 82, pro=o.PRO_LN_PRELU, pro_gamma=rnd(256, seed=54), pro_beta=rnd(256, seed=55),
        pro_alpha=torch.tensor([0.1], device=DEV))
    add(25728, 512, 128, 84, pro=o.PRO_ROWLN, pro_gamma=rnd(128, seed=56), pro_beta=rnd(128, seed=57),
        pro_stats=torch.rand(25728, 2, device=DEV))
    x2 = rnd(700, H, seed=62)
    idx2 = torch.randint(0, 700, (17000,), device=DEV, dtype=torch.int32)
    e2 = rnd(17000, H, seed=63)
    add(17000, 192, 2 * H, 86, segs=[o.seg(x2, rmap=o.rowmap(idx=idx2)), o.seg(e2)])
    jobs[-1]["keep"] = (x2, idx2, e2)
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

# ---------------------------------------------------------------------------------------------------------------------
# nodes with more incoming edges than a message-GEMM tile holds (48): periodic neighbour lists at r_max = 4 A give them
# (`utils.py:267`); scatter_mean / scatter_sum of `DOSTransformer_phonon.py:209` / `DOSTransformer.py:187` have no limit
# ---------------------------------------------------------------------------------------------------------------------
def _fatten(c, node, extra, seed):
    """`extra` more edges into `node` of crystal dict c (sources uniform over the real atoms, fresh edge features)."""
    g = torch.Generator().manual_seed(seed)
    n_real = int(c["x"].shape[0]) - (1 if "edge_attr" in c else 0)           # eDOS: the last node is the phantom node
    src = torch.randint(0, n_real, (extra,), generator=g)
    ei = torch.cat([c["edge_index"], torch.stack([src, torch.full((extra,), node, dtype=torch.int64)])], 1)
    out = dict(c)
    out["edge_index"] = ei
    if "edge_vec" in c:
        v = (torch.rand(extra, 3, generator=g, dtype=torch.float64) * 2 - 1) * 2.3
        out["edge_vec"] = torch.cat([c["edge_vec"], v.to(c["edge_vec"].dtype)], 0)
    else:
        d = torch.rand(extra, generator=g, dtype=torch.float64) * 7.0 + 1.0
        mu = torch.arange(41, dtype=torch.float64) * 0.2
        out["edge_attr"] = torch.cat([c["edge_attr"], torch.exp(-((d[:, None] - mu[None, :]) ** 2) / 0.04).to(c["edge_attr"].dtype)], 0)
    return out

def _fat_crystals(kind, B, seed, dtype):
    from dostransformer_amd import synth
    cs = synth.phonon_crystals(B, seed, dtype) if kind == "phonon" else synth.edos_crystals(B, seed, dtype)
    cs[0] = _fatten(cs[0], 0, 45, 1)            # in-degree ~ 60: one full chunk + a remainder that shares its tile
    cs[1] = _fatten(cs[1], 1, 185, 2)           # ~ 200: four full chunks + remainder
    cs[2] = _fatten(_fatten(cs[2], 0, 96 - int((cs[2]["edge_index"][1] == 0).sum()), 3), 1, 70, 4)   # exactly 96 (two full chunks,
    return cs                                   # no remainder) next to another over-full node

class _Hog:
    """Keeps a second stream busy streaming 2 x 512 MiB buffers through HBM (every XCD's L2 is being thrashed and the
    memory channels are loaded while the kernels under test publish / read back their partial results)."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.a = torch.empty(128 << 20, device=DEV)
        self.b = torch.empty(128 << 20, device=DEV)
        self.n = 0

    def feed(self, k=1):
        with torch.cuda.stream(self.stream):
            for _ in range(k):
                self.b.copy_(self.a)
                self.n += 1

class _FakeDist:
    """Two-rank stand-in whose collectives are no-ops (the sums of a rank with an identical twin would double everything,
    which this test does not look at): what is under test is the n_global / shard-size logic of Trainer.step_dataset."""
    world, rank, staged = 2, 0, False

    def __init__(self, sizes):
        self.sizes, self.calls = sizes, 0

    def min_max(self, v):
        self.calls += 1
        return self.sizes if self.sizes is not None else (v, v)

    def all_reduce_sse(self, t):
        pass

    def all_reduce_grads(self, t):
        pass

    def all_reduce_grads_async(self, t):
        from dostransformer_amd.dist import _Done
        return _Done()

def _sliver_case(o, _lib, Gemm, M, N, K, mapped, res):
    rows_a = 2 * M if mapped else M
    a, w = rnd(rows_a, K, seed=1), rnd(K, N, seed=2)
    r = rnd(M, N, seed=3) if res else None
    out = torch.full((M, N), float("nan"), device=DEV)
    B = max(M // 51, 1)
    rm = o.rowmap(d=B, m=2 * B, c=1, off=B) if (mapped and M % 51 == 0) else None
    if mapped and rm is None:
        pytest.skip("row-map case needs M = 51 * B")
    o.gemm(M, N, [o.seg(a, rmap=rm)], w, out, w_layout=1, res=r)
    torch.cuda.synchronize()
    idx = torch.arange(M, device=DEV)
    if rm is not None:
        idx = (idx // B) * (2 * B) + (idx % B) + B
    ref = a.double()[idx] @ w.double() + (r.double() if res else 0.0)
    assert not torch.isnan(out).any() and err(out, ref) < TOL
    g = Gemm()
    g.M, g.N, g.K, g.nseg = M, N, K, 1
    g.a[0] = o.seg(a, rmap=rm)
    g.w, g.ldw, g.w_layout = w.data_ptr(), N, 1
    g.out, g.ldo, g.out_map, g.res_map = out.data_ptr(), N, o.ident(), o.ident()
    buf = C.create_string_buffer(96)
    _lib.load().dosx_gemm_kernel_name(C.byref(g), buf, 96)
    small = 2.0 * M * N * K <= 2e9
    assert (buf.value.decode() == "sliver_gemm_kernel") == small, buf.value

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

def _node_block(M, H, seed, dev="cuda:0"):
    gen = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=gen)).to(dev)
    x, agg = r(M, H), r(M, H)
    W = dict(w1=r(2 * H, 2 * H) / (2 * H) ** 0.5, b1=0.1 * r(2 * H), g=1 + 0.1 * r(2 * H), b=0.1 * r(2 * H),
             al=torch.tensor([0.25], device=dev), w2=r(H, 2 * H) / (2 * H) ** 0.5, b2=0.1 * r(H))
    return x, agg, W, r
