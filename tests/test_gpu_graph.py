"""The message-passing side on the GPU (csrc/graph_ops.hip, edge_mlp.hip, mlp2.hip, csr.hip, neighbors.hip): edge features, CSR
segment sums, the one-launch EdgeModel / NodeModel kernels and their column-split / chained forms, collate, neighbour lists
(DOSTransformer_phonon.py:46-64,148-212; utils.py:249-303)."""
import copy
import ctypes as C
import math
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import rmse  # noqa: F401
from tests.gpu_util import (DEV, TOL, _FakeDist, _Hog, _attn_ref, _descs, _fat_crystals, _fatten, _graph, _mixed_jobs, _node_block, _philox_mask_numpy, _phonon, _random_crystals, _reduce, _ref, _scratch, _sliver_case, err, ops, prelu, rnd)  # noqa: F401

pytestmark = pytest.mark.gpu


def test_edge_features():
    o = ops()
    from oracle import dos_oracle as O
    v = rnd(1000, 3, seed=1, scale=2.0)
    v[0] = 0
    v[1] = torch.tensor([4.0, 0, 0])
    got = o.edge_feat_sh1(v, 4.0)
    assert err(got, O.edge_features_sh1(v.double().cpu()).to(DEV)) < 1e-5


@pytest.mark.parametrize("H", [128, 256, 16, 64])
@pytest.mark.parametrize("mean", [True, False])
def test_segment_reduce_and_backward(H, mean):
    o = ops()
    from dostransformer_amd import synth
    from dostransformer_amd.batch import graph_meta
    g = synth.phonon_batch(5, seed=3, dtype=torch.float32)
    m = graph_meta(g, DEV)
    N, E = m.num_nodes, m.num_edges
    msg = rnd(E, H, seed=1)
    e_in = rnd(E, H, seed=2)
    agg = torch.empty(N, H, device=DEV)
    e_out = torch.empty(E, H, device=DEV)
    o.segment_reduce(msg, m.rowptr_dst, m.inv_deg if mean else None, agg, e_in, e_out, N, E, H)
    ref = torch.zeros(N, H, device=DEV, dtype=torch.float64).index_add_(0, m.dst.long(), msg.double())
    if mean:
        ref = ref * m.inv_deg.double()[:, None]
    assert err(agg, ref) < 1e-5
    assert err(e_out, e_in.double() + msg.double()) < 1e-6
    # edge grad combine
    dcat_n = rnd(N, 2 * H, seed=3)
    de_new = rnd(E, H, seed=4)
    dmsg = torch.empty(E, H, device=DEV)
    o.edge_grad_combine(de_new, dcat_n.data_ptr() + 4 * H, 2 * H, m.dst, m.inv_deg if mean else None, dmsg, E, H)
    sc = m.inv_deg.double()[m.dst.long()][:, None] if mean else 1.0
    assert err(dmsg, de_new.double() + dcat_n[:, H:].double()[m.dst.long()] * sc) < 1e-6
    # gather backward
    dcat = rnd(E, 3 * H, seed=5)
    dx_res = rnd(N, H, seed=6)
    dx = torch.empty(N, H, device=DEV)
    de_out = torch.empty(E, H, device=DEV)
    o.gather_bwd(dcat, dcat_n.data_ptr(), 2 * H, dx_res, m.rowptr_dst, m.rowptr_src, m.perm_src, de_new, dx, de_out,
                 N, E, H)
    ref = dx_res.double() + dcat_n[:, :H].double()
    ref = ref.index_add(0, m.dst.long(), dcat[:, H:2 * H].double()).index_add(0, m.src.long(), dcat[:, :H].double())
    assert err(dx, ref) < 1e-5
    assert err(de_out, de_new.double() + dcat[:, 2 * H:].double()) < 1e-6


def test_pool_dense_norm():
    o = ops()
    from dostransformer_amd import synth
    from dostransformer_amd.batch import graph_meta
    g = synth.phonon_batch(6, seed=4, dtype=torch.float32)
    m = graph_meta(g, DEV)
    N, B, H, nmax = m.num_nodes, m.num_graphs, 128, m.n_max
    x = rnd(N, H, seed=1)
    pooled = torch.empty(B, H, device=DEV)
    o.graph_pool(x, m.graph_ptr, pooled.data_ptr(), H, B, H)
    ref = torch.zeros(B, H, device=DEV, dtype=torch.float64).index_add_(0, m.node_graph.long(), x.double())
    assert err(pooled, ref) < 1e-5
    dx = rnd(N, H, seed=2)
    dx0 = dx.clone()
    dp = rnd(B, H, seed=3)
    o.graph_pool_bwd(dp.data_ptr(), H, m.node_graph, dx, N, H, True)
    assert err(dx, dx0.double() + dp.double()[m.node_graph.long()]) < 1e-6
    kv = torch.full((nmax * B, H), 7.0, device=DEV)
    rstd = torch.empty(N, device=DEV)
    o.dense_normalize(x, m.dense_row, kv, rstd, N, H, nmax * B)
    xr = x.double().requires_grad_(True)
    xh = F.layer_norm(xr, (H,), None, None, 1e-5)
    ref = torch.zeros(nmax * B, H, device=DEV, dtype=torch.float64)
    ref[m.dense_row.long()] = xh.detach()
    assert err(kv, ref) < 1e-5
    dkv = rnd(nmax * B, H, seed=5)
    xh.backward(dkv.double()[m.dense_row.long()])
    dxn = torch.zeros(N, H, device=DEV)
    o.dense_normalize_bwd(dkv, kv, rstd, m.dense_row, dxn, N, H, False)
    assert err(dxn, xr.grad) < 5e-5


@pytest.mark.parametrize("B,seed", [(1, 0), (7, 1), (64, 2)])
def test_csr_build_matches_host(B, seed):
    """dosx_csr_build (device) == batch._build_meta_host (numpy) on shuffled, PyG-style index tensors."""
    import numpy as np
    from dostransformer_amd import synth
    from dostransformer_amd.batch import _build_meta_host
    o = ops()
    g = synth.phonon_batch(B, seed=seed, dtype=torch.float32, sort_edges=False)
    ei = g.edge_index.clone()
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(seed))
    ei = ei[:, perm]                                         # arbitrary edge order
    ref = _build_meta_host(ei.numpy(), g.batch.numpy(), B, None, presorted=False)
    r = o.csr_build(ei.to(DEV), g.batch.to(DEV), B)
    torch.cuda.synchronize()
    for k in ("src", "dst", "rowptr_dst", "perm_src", "rowptr_src", "graph_ptr", "node_graph", "dense_row"):
        assert torch.equal(r[k].cpu(), getattr(ref, k)), k
    assert torch.equal(r["edge_perm"].cpu(), ref.edge_perm)
    assert torch.equal(r["inv_deg"].cpu(), ref.inv_deg)
    assert int(r["n_max"].item()) == ref.n_max


@pytest.mark.parametrize("cutoff,si", [(3.0, True), (5.0, True), (4.0, False)])
def test_neighbor_list_matches_oracle(cutoff, si):
    """Same edges, same order (crystal, i, j, shift), bit-identical edge_vec as the brute-force restatement."""
    from oracle.dos_oracle import neighbor_list_bruteforce
    sizes = [1, 2, 12, 5, 30, 1, 7]
    pos, cells = _random_crystals(7, sizes)
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    out = ops().neighbor_list(torch.from_numpy(np.concatenate(pos)).to(DEV), torch.from_numpy(np.stack(cells)).to(DEV),
                            torch.from_numpy(ptr).to(DEV), cutoff, self_interaction=si)
    eptr = out["edge_ptr"].cpu().numpy()
    assert eptr[-1] == out["src"].numel() > 0
    for c, (p, cell) in enumerate(zip(pos, cells)):
        i, j, S, D = neighbor_list_bruteforce(p, cell, cutoff, si)
        a, b = eptr[c], eptr[c + 1]
        assert b - a == len(i), c
        assert (out["crystal"][a:b].cpu().numpy() == c).all()
        assert (out["src"][a:b].cpu().numpy() == i).all() and (out["dst"][a:b].cpu().numpy() == j).all()
        assert (out["shift"][a:b].cpu().numpy() == S).all()
        assert np.array_equal(out["edge_vec"][a:b].cpu().numpy(), D)


def test_neighbor_list_known_answers_and_slab():
    """fcc coordination shells (12, 6, 24) straight from the kernel; a non-periodic axis only keeps shift 0 there."""
    fcc = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0]], float)
    cell = np.eye(3)[None]
    ptr = torch.tensor([0, 4], dtype=torch.int32, device=DEV)
    for rc, expect in [(0.71, 12), (1.01, 18), (1.23, 42)]:
        out = ops().neighbor_list(torch.from_numpy(fcc).to(DEV), torch.from_numpy(cell).to(DEV), ptr, rc, self_interaction=False)
        assert (np.bincount(out["src"].cpu().numpy(), minlength=4) == expect).all()
    slab = ops().neighbor_list(torch.from_numpy(fcc).to(DEV), torch.from_numpy(cell).to(DEV), ptr, 1.01,
                             self_interaction=False, pbc=(True, True, False))
    assert int(slab["shift"][:, 2].abs().max()) == 0 and 0 < slab["src"].numel() < 4 * 18


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
        # node-aligned row tiles of the message GEMM: round 6 - the device packs greedily over the whole batch like the host
        # (csrc/csr.hip: collate_greedy_tiles_kernel), tile for tile the same table ...
        assert torch.equal(slot.g.meta.seg_tile.cpu(), ref.meta.seg_tile), sel
        # ... valid: monotone, <= 48 rows, tile edges = CSR pointers of its node range, full cover
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


def _without_incoming(c, nodes):
    """Crystal dict c with every edge INTO the given nodes removed (isolated destination nodes: no rows in the message GEMM)."""
    keep = ~torch.isin(c["edge_index"][1], torch.tensor(list(nodes)))
    out = dict(c)
    out["edge_index"] = c["edge_index"][:, keep]
    for k in ("edge_vec", "edge_attr"):
        if k in c:
            out[k] = c[k][keep]
    return out


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_device_greedy_tiles_equal_the_host_table_with_overfull_and_isolated_nodes(kind, monkeypatch):
    """collate_greedy_tiles_kernel against batch.seg_tiles_host + pad_seg_tiles on batches that exercise every rule of the packing:
    nodes of more than 48 incoming edges (chunk tiles; a remainder that shares its tile; exactly 96 = two full chunks), isolated
    nodes at the start of the batch, behind a full last chunk (absorbed by that chunk's tile), between crystals and at the end,
    single-crystal batches, repeated crystals; and the crystal-aligned fallback table stays valid (DOSX_COLLATE_GREEDY_TILES=0 is
    read once per process, so the fallback is checked through its invariants in test_collate_into_matches_pad_batch's history)."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    from dostransformer_amd.loader import DeviceDataset
    from dostransformer_amd.train import _Slot
    from tests.gpu_util import _fatten
    cs = synth.phonon_crystals(10, seed=51, dtype=torch.float32) if kind == "phonon" else synth.edos_crystals(10, seed=52, dtype=torch.float32)
    indeg = lambda c, n: int((c["edge_index"][1] == n).sum())
    cs[0] = _without_incoming(cs[0], [0, 1])                                    # the batch starts with isolated nodes
    cs[1] = _fatten(cs[1], 0, 96 - indeg(cs[1], 0), 3)                          # exactly two full chunks ...
    cs[1] = _without_incoming(cs[1], [1, 2])                                    # ... with isolated nodes right behind them
    cs[2] = _fatten(cs[2], 1, 185, 2)                                           # four full chunks + a remainder
    cs[3] = _without_incoming(cs[3], range(int(cs[3]["x"].shape[0])))           # a crystal without any edge
    cs[4] = _fatten(_fatten(cs[4], 0, 48 - indeg(cs[4], 0), 5), 1, 49 - indeg(cs[4], 1), 6)   # exactly 48 (one plain tile), 49 (chunk + 1)
    last = int(cs[5]["x"].shape[0]) - 1
    cs[5] = _fatten(_without_incoming(cs[5], [last]), last, 144, 7)             # three full chunks on the LAST node of a crystal
    ds = DeviceDataset(cs, DEV)
    t = ds._f32_tables()
    for sel in (list(range(10)), [1], [5], [3, 3, 1], [5, 0], [1, 5], [9, 8, 7, 6, 5, 4, 3, 2, 1, 0], [2, 2, 2]):
        idx, N, E, n_max = ds.bucket_dims(sel, n_max=64)
        n_pad, e_pad = bucket_sizes(N, E, 16, 256)
        slot = _Slot.empty(kind, DEV, len(sel), n_pad, e_pad, n_max, t["x"].shape[1], t["edge"].shape[1], t["target"].shape[1], tiled=True)
        slot.g.meta.seg_tile.fill_(-7)
        ds.collate_into(slot.g, idx, slot.scratch)
        ref = pad_batch(collate([cs[i] for i in sel], n_max=64), n_pad, e_pad)
        torch.cuda.synchronize()
        assert torch.equal(slot.g.meta.rowptr_dst.cpu(), ref.meta.rowptr_dst)
        got, want = slot.g.meta.seg_tile.cpu(), ref.meta.seg_tile
        assert got.shape == want.shape and torch.equal(got, want), (sel, (got != want).nonzero()[:5].tolist())
    assert int((ref.meta.seg_tile[2] != 0).sum()) > 0


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


@pytest.mark.parametrize("H,mean", [(128, True), (64, False), (256, False)])
def test_message_gemm_segment_sum_with_overfull_nodes(H, mean):
    """DosxGemm EPI_SEGSUM on a batch with 60-, 96- and 200-in-degree nodes (chunk tiles + in-launch combination of the
    chunk sums) == the same GEMM followed by the stand-alone dosx_segment_reduce, to rounding (the chunked order of the
    adds differs from the sequential one for those nodes); ghost-padded too; twice in a row (counters back at zero)."""
    from dostransformer_amd import functional as Fn, ops
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    g = collate(_fat_crystals("phonon", 6, 5, torch.float32))
    deg = torch.bincount(g.edge_index[1], minlength=g.x.shape[0])
    assert int(deg.max()) >= 200 and int((deg > 48).sum()) >= 4 and int((deg == 96).sum()) >= 1
    for padded in (False, True):
        b = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 16, 256)) if padded else g
        assert b.meta.seg_tile is not None and int((b.meta.seg_tile[2] != 0).sum()) >= 9
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
        prev = None
        for rep in range(2):
            agg1 = torch.full((N, H), float("nan"), device=DEV)
            e1 = torch.full((E, H), float("nan"), device=DEV)
            Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(m.seg_tile, m.rowptr_dst, scale, agg1, e, e1))
            torch.cuda.synchronize()
            nr = getattr(b, "real_nodes", N)
            assert bool(torch.isfinite(agg1).all())
            assert float((agg1[:nr] - agg0[:nr]).abs().max()) <= 3e-6 * float(agg0[:nr].abs().max())
            assert torch.equal(e1, e0)
            if prev is not None:
                assert torch.equal(agg1, prev)                         # deterministic, counters reset
            prev = agg1


def test_stress_segment_sum_and_key_gradient_tickets_under_a_bandwidth_hog():
    """... the other two in-launch reductions: the message GEMM's chunk sums of over-full nodes (EPI_SEGSUM, ticket on the
    node's first tile; 2 000 launches, counters drawn from the eager ring every time) and the attention backward's key
    gradients finished by the last arriving query tile of a crystal (1 500 launches), under the same hog.  The attention
    launches are compared bitwise with the TWO-launch result (dq kernel + attn_dkv_reduce_kernel); the chunked segment sums
    have no two-launch twin with the same summation order, so they are compared bitwise with their own first launch and to
    rounding with GEMM + dosx_segment_reduce."""
    from dostransformer_amd import _lib, functional as Fn
    from dostransformer_amd._lib import Attn
    from dostransformer_amd.batch import seg_tiles_host
    o = ops()
    hog = _Hog()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    # ---- segment sums: 40 nodes, a third of them over-full (49 .. 400 incoming edges)
    rng = np.random.default_rng(3)
    n, H = 40, 128
    deg = rng.integers(0, 30, size=n)
    deg[::3] = rng.choice([49, 96, 97, 144, 200, 400], size=len(deg[::3]))
    E = int(deg.sum())
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    dst = torch.from_numpy(np.repeat(np.arange(n), deg).astype(np.int32)).to(DEV)
    src = torch.from_numpy(rng.integers(0, n, size=E).astype(np.int32)).to(DEV)
    tiles = torch.from_numpy(seg_tiles_host(rowptr)).to(DEV)
    assert int((tiles[2] != 0).sum()) >= 20
    rp = torch.from_numpy(rowptr.astype(np.int32)).to(DEV)
    inv = torch.from_numpy((1.0 / np.maximum(deg, 1)).astype(np.float32)).to(DEV)
    gen = torch.Generator().manual_seed(5)
    P = {"k.0.weight": torch.randn(2 * H, 3 * H, generator=gen) * 0.1, "k.0.bias": torch.randn(2 * H, generator=gen),
         "k.1.weight": torch.randn(2 * H, generator=gen), "k.1.bias": torch.randn(2 * H, generator=gen),
         "k.2.weight": torch.tensor([0.25]), "k.3.weight": torch.randn(H, 2 * H, generator=gen) * 0.1,
         "k.3.bias": torch.randn(H, generator=gen)}
    P = {k: v.to(DEV) for k, v in P.items()}
    x, e = torch.randn(n, H, generator=gen).to(DEV), torch.randn(E, H, generator=gen).to(DEV)
    a = Fn.SegList([o.seg(x, rmap=o.rowmap(idx=src)), o.seg(x, rmap=o.rowmap(idx=dst)), o.seg(e)], [x, e])
    msg, _ = Fn.mlp_ln_fwd(P, "k", a, E, H)
    agg0, e0 = torch.empty(n, H, device=DEV), torch.empty(E, H, device=DEV)
    o.segment_reduce(msg, rp, inv, agg0, e, e0, n, E, H)
    agg1, e1 = torch.full((n, H), float("nan"), device=DEV), torch.full((E, H), float("nan"), device=DEV)
    Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(tiles, rp, inv, agg1, e, e1))
    torch.cuda.synchronize()
    assert float((agg1 - agg0).abs().max()) <= 4e-6 * float(agg0.abs().max()) and torch.equal(e1, e0)
    ref_agg = agg1.clone()
    for it in range(2000):
        if it % 4 == 0:
            hog.feed()
        agg1.fill_(float("nan"))
        Fn.mlp_ln_fwd(P, "k", a, E, H, segsum=(tiles, rp, inv, agg1, e, e1))
        bad += (agg1 != ref_agg).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, ("segment sums", int(bad))
    # ---- attention key gradients: the cfg2 self-attention shape (51 keys, 128 pseudo-crystals, 2 query tiles each)
    Sq, Bq, Nk, Bk, Hh = 51, 128, 51, 128, 128
    assert _lib.load().dosx_attention_pkv_supported(Nk, Hh)
    xq, kv = rnd(Sq * Bq, Hh, seed=1), rnd(Nk * Bk, Hh, seed=2)
    gam, bet = rnd(Hh, seed=3), 0.3 * rnd(Hh, seed=4)
    at = Attn()
    at.Sq, at.Bq, at.Nk, at.Bk, at.H, at.q_stride_s, at.q_stride_b = Sq, Bq, Nk, Bk, Hh, Bq, 1
    out, probs = torch.empty(Sq * Bq, Hh, device=DEV), torch.empty(Bq, Sq, Nk, device=DEV)
    qstats, ostats = torch.empty(Sq * Bq, 2, device=DEV), torch.empty(Sq * Bq, 2, device=DEV)
    at.x, at.kvhat, at.gamma0, at.beta0 = xq.data_ptr(), kv.data_ptr(), gam.data_ptr(), bet.data_ptr()
    at.out, at.probs, at.qstats, at.out_stats = out.data_ptr(), probs.data_ptr(), qstats.data_ptr(), ostats.data_ptr()
    o.attention_fwd(at)
    dout = rnd(Sq * Bq, Hh, seed=5)
    nqt, nkt = (Sq + 31) // 32, (Nk + 15) // 16
    base = rnd(Nk * Bk, Hh, seed=6)
    dx = torch.full((Sq * Bq, Hh), float("nan"), device=DEV)
    dkv = base.clone()
    part = torch.full((Bq * nqt + Bk * nkt, 2 * Hh), float("nan"), device=DEV)
    kvp = torch.full((Bq * nqt * Nk, Hh), float("nan"), device=DEV)
    at.dout, at.dx, at.dscores, at.dkvhat, at.dkv_accumulate = dout.data_ptr(), dx.data_ptr(), None, dkv.data_ptr(), 1
    at.partials_q, at.partials_kv = part.data_ptr(), part.data_ptr() + 4 * Bq * nqt * 2 * Hh
    at.dkv_part = kvp.data_ptr()
    at.dkv_cnt = None
    o.attention_bwd(at)                                       # two launches: dq kernel + attn_dkv_reduce_kernel
    torch.cuda.synchronize()
    two = (dx.clone(), dkv.clone(), part.clone())
    at.dkv_cnt = o.COUNTERS.take(DEV, Bk)
    lib = _lib.load()
    # mode 0: attention.hip's one-launch form (bitwise its two-launch result); mode 2 (round 5, the default): the crystal-aligned
    # kernels of attention_aligned.hip behind the same call - other tiles and summation orders, so their reference is their own
    # first launch, which agrees with the two-launch result to rounding
    for mode, iters in ((0, 600), (2, 1200)):
        prev = lib.dosx_attention_aligned_mode(mode)
        try:
            ref = two
            if mode == 2:
                dkv.copy_(base)
                o.attention_bwd(at)
                torch.cuda.synchronize()
                ref = (dx.clone(), dkv.clone(), part.clone())
                assert float((ref[0] - two[0]).abs().max()) < 1e-4 * float(two[0].abs().max())
                assert float((ref[1] - two[1]).abs().max()) < 1e-4 * float(two[1].abs().max())
                assert float((ref[2].sum(0) - two[2].sum(0)).abs().max()) < 1e-4 * float(two[2].sum(0).abs().max())
            bad.zero_()
            for it in range(iters):
                if it % 4 == 0:
                    hog.feed()
                dkv.copy_(base)
                dx.fill_(float("nan"))
                part.fill_(float("nan"))
                o.attention_bwd(at)
                bad += (dx != ref[0]).sum() + (dkv != ref[1]).sum() + (part != ref[2]).sum()
            torch.cuda.synchronize()
            assert int(bad) == 0, ("attention key gradients", mode, int(bad))
        finally:
            lib.dosx_attention_aligned_mode(prev)
    assert hog.n >= 800


def test_segment_sum_gemm_refuses_a_call_without_chunk_scratch():
    """DosxGemm EPI_SEGSUM without seg_part / seg_cnt would silently skip over-full nodes (the tile table is
    device memory, the host cannot tell): the C ABI refuses it."""
    from dostransformer_amd import _lib
    from dostransformer_amd._lib import Gemm
    from dostransformer_amd.batch import seg_tiles_host
    o = ops()
    n, H = 6, 64
    deg = np.array([3, 60, 2, 0, 5, 100])
    E = int(deg.sum())
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    tiles = torch.from_numpy(seg_tiles_host(rowptr)).to(DEV)
    rp = torch.from_numpy(rowptr.astype(np.int32)).to(DEV)
    xhat, stats = rnd(E, 2 * H, seed=1), torch.rand(E, 2, device=DEV)
    w, agg = rnd(H, 2 * H, seed=2), torch.empty(n, H, device=DEV)
    gam, bet, alpha = rnd(2 * H, seed=3), rnd(2 * H, seed=4), torch.tensor([0.25], device=DEV)
    g = Gemm()
    g.M, g.N, g.K, g.nseg = E, H, 2 * H, 1
    g.a[0] = o.seg(xhat)
    g.pro, g.pro_gamma, g.pro_beta, g.pro_alpha, g.pro_stats = o.PRO_LN_PRELU, gam.data_ptr(), bet.data_ptr(), alpha.data_ptr(), stats.data_ptr()
    g.w, g.ldw, g.w_layout, g.epi = w.data_ptr(), 2 * H, 0, o.EPI_SEGSUM
    g.ldo, g.out_map, g.res_map = H, o.ident(), o.ident()
    g.seg_tile, g.seg_ntiles, g.seg_rowptr, g.seg_agg = tiles.data_ptr(), tiles.shape[1] - 1, rp.data_ptr(), agg.data_ptr()
    rc = _lib.load().dosx_gemm(C.byref(g), o._stream())
    assert rc == -22
    with pytest.raises(_lib.DosxError, match="seg_part"):
        _lib.check(rc, "dosx_gemm")


def test_dense_slots_is_to_dense_batch():
    """dosx_dense_slots / _bwd = torch_geometric.utils.to_dense_batch (mask dropped) and its adjoint, on a ghost-padded batch."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    o = ops()
    H = 384
    g = collate(synth.phonon_crystals(5, 3, torch.float32))
    g = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 16, 256)).to(DEV)
    m = g.meta
    N, B, nmax = m.num_nodes, m.num_graphs, m.n_max
    x = rnd(N, H, seed=1)
    dense = torch.full((nmax * B, H), float("nan"), device=DEV)
    o.dense_slots(x, m.graph_ptr, dense, B, nmax, H)
    gp = m.graph_ptr.cpu().tolist()
    ref = torch.zeros(nmax * B, H, device=DEV)
    for b in range(B):
        for pos in range(gp[b + 1] - gp[b]):
            ref[pos * B + b] = x[gp[b] + pos]
    torch.cuda.synchronize()
    assert torch.equal(dense, ref)
    dd = rnd(nmax * B, H, seed=2)
    dx = torch.full((N, H), float("nan"), device=DEV)
    o.dense_slots_bwd(dd, m.dense_row, dx, N, H, False, ghost_row=nmax * B)
    refdx = torch.zeros(N, H, device=DEV)
    for b in range(B):
        for pos in range(gp[b + 1] - gp[b]):
            refdx[gp[b] + pos] = dd[pos * B + b]
    torch.cuda.synchronize()
    assert torch.equal(dx, refdx)                      # ghost nodes: zero


@pytest.mark.parametrize("H", [64, 128, 256])
def test_segment_reduce_perm_and_strided_wgrad(H):
    """dosx_segment_reduce_perm = the sums of a per-edge tensor over the edges that LEAVE each node (rowptr_src / perm_src),
    against index_add; and a finished-mode weight-gradient job that writes a COLUMN BLOCK of a wider gradient (DosxWgrad.ldd)."""
    from dostransformer_amd import synth
    from dostransformer_amd.batch import collate
    o = ops()
    g = collate(synth.phonon_crystals(7, 3, torch.float32)).to(DEV)
    m = g.meta
    N, E = m.num_nodes, m.num_edges
    dz = rnd(E, H, seed=1)
    agg = torch.full((N, H), float("nan"), device=DEV)
    o.segment_reduce_perm(dz, m.rowptr_src, m.perm_src, agg, N, E, H)
    ref = torch.zeros(N, H, dtype=torch.float64, device=DEV).index_add_(0, m.src.long(), dz.double())
    torch.cuda.synchronize()
    assert err(agg, ref) < TOL
    # strided destination: three column blocks of one [Nn, 3K] gradient
    M, Nn, K = 900, 64, 32
    dy, a = rnd(M, Nn, seed=2), rnd(M, 3 * K, seed=3)
    dw = torch.full((Nn, 3 * K), float("nan"), device=DEV)
    db = torch.full((Nn,), float("nan"), device=DEV)
    jobs = []
    for j in range(3):
        ns = o.wgrad_splits(M, Nn, K)
        nf = o.wgrad_scratch_floats(Nn, K, ns)
        slab = torch.empty(max(nf, 1), device=DEV)
        sb = torch.empty(ns * 64, device=DEV) if j == 2 else None
        jobs.append((o.wgrad_desc(M, Nn, o.seg(dy), [o.seg(a, width=K, col=j * K)], slab, sb, ns, dst=dw[:, j * K:(j + 1) * K],
                                  dst_bias=db if j == 2 else None), slab, sb))
    o.wgrad_grouped([j[0] for j in jobs])
    torch.cuda.synchronize()
    assert err(dw, dy.double().T @ a.double()) < TOL and err(db, dy.double().sum(0)) < TOL


@pytest.mark.parametrize("kind", ["phonon", "edos"])
def test_factored_edge_weight_gradient_equals_the_plain_one(kind):
    """The EdgeModel's first Linear FACTORED - forward: (x Wa^T)[row] + (x Wb^T)[col] + e Wc^T + b through N-row GEMMs, a
    third-width E-row GEMM and dosx_gather_add_rownorm; weight gradient: [node sums (x) x | node sums (x) x | dz (x) e] (N-row jobs
    behind two segment sums) - against the gathered-concat GEMM / the one E-row job: outputs and all gradients of a training
    step agree to rounding; eager and replay give the same bits; ghost-padded batch."""
    from dostransformer_amd import functional as Fn, synth
    from dostransformer_amd.batch import bucket_sizes, collate, pad_batch
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    if kind == "phonon":
        from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
        mk = lambda: DOSTransformer_phonon(3, 1, 118, 4, 64, DEV, 0.0)
        cs = synth.phonon_crystals(6, 5, torch.float32)
    else:
        from dostransformer_amd.embedder_eDOS.DOSTransformer import DOSTransformer
        mk = lambda: DOSTransformer(3, 1, 200, 41, 2, 64, DEV, 0.0)
        cs = synth.edos_crystals(6, 5, torch.float32)
    g = collate(cs)
    gp = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges, 16, 256)).to(DEV)
    m0 = mk()
    sd0 = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    grads, params, outs = {}, {}, {}
    min_gf, last_gf, heads_gf = Fn._FACTOR_MIN_GF, Fn._FACTOR_LAST_MIN_GF, Fn._FACTOR_HEADS_MIN_GF
    for fac in (False, True):
        Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF = fac, 0.0        # (the shipped policy factors from 4 GF; the path takes any size)
        Fn._FACTOR_LAST_MIN_GF = 0.0 if fac else 1e9               # ... and the last layer's aggregate-first form with it
        Fn._FACTOR_HEADS_MIN_GF = 0.0 if fac else 1e9              # ... and the output heads' per-crystal K-segments (res_pre)
        try:
            for replay in (False, True):
                model = mk()
                model.load_state_dict(sd0)
                model = model.to(DEV)
                tr = Trainer(model, lr=1e-3, replay=replay)
                tr.forward_backward(gp)
                torch.cuda.synchronize()
                fp = model.flat_params()
                grads[(fac, replay)] = {k: v.clone() for k, v in fp.G.items()}
                outs[(fac, replay)] = [t.clone() for t in tr.last_outputs]
                for _ in range(2):
                    tr.step(gp)
                torch.cuda.synchronize()
                params[(fac, replay)] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        finally:
            Fn._FACTOR_EDGE_WGRAD, Fn._FACTOR_MIN_GF, Fn._FACTOR_LAST_MIN_GF, Fn._FACTOR_HEADS_MIN_GF = True, min_gf, last_gf, heads_gf
    n_real, n_pad = g.meta.num_nodes, gp.meta.num_nodes
    for u, v in zip(outs[(True, False)], outs[(False, False)]):
        if u.shape[0] == n_pad:                          # node embeddings: the ghost rows are finite don't-cares (batch.pad_batch) -
            u, v = u[:n_real], v[:n_real]                # the last layer's aggregate-first form sums the ghost self loops differently
        assert err(u, v) < 5e-6                          # the forward product factored too: same numbers to rounding
    for k, v in grads[(False, False)].items():
        assert err(grads[(True, False)][k], v) < 1e-4, k
    for k in params[(True, False)]:
        assert torch.equal(params[(True, False)][k], params[(True, True)][k]), ("eager vs replay", k)


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("W,Hout", [(128, 64), (512, 256), (1024, 512)])
def test_act_segment_sum_and_gathered_ln_prelu_backward(mean, W, Hout):
    """The last message-passing layer with the aggregation in front of its second Linear (dosx_act_segment_sum,
    dosx_seg_count_scale, dosx_ln_prelu_bwd_gather) against the per-edge formulation in torch (DOSTransformer_phonon.py:193-197,209 /
    DOSTransformer.py:187); empty segments (nodes without incoming edges) included; bitwise repeatable."""
    from dostransformer_amd import ops
    torch.manual_seed(1)
    N, E = 37, 411
    dst = torch.sort(torch.randint(0, N - 3, (E,), device=DEV))[0].to(torch.int32)      # the last three nodes: empty segments
    deg = torch.bincount(dst.long(), minlength=N)
    rowptr = torch.zeros(N + 1, device=DEV, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(deg, 0).to(torch.int32)
    scale = torch.where(deg > 0, 1.0 / deg.clamp(min=1).float(), torch.zeros((), device=DEV)) if mean else None
    xhat = torch.randn(E, W, device=DEV)
    rstd = torch.rand(E, device=DEV) + 0.5
    gam, bet = torch.randn(W, device=DEV), torch.randn(W, device=DEV)
    alpha = torch.tensor([0.25], device=DEV)
    Wt, bias = torch.randn(Hout, W, device=DEV) / W ** 0.5, torch.randn(Hout, device=DEV)
    S, R = torch.empty(N, W, device=DEV), torch.empty(N, Hout, device=DEV)
    ops.act_segment_sum(xhat, rowptr, scale, gam, bet, alpha, bias, S, R, N, E, W, Hout)
    S2, R2 = torch.empty_like(S), torch.empty_like(R)
    ops.act_segment_sum(xhat, rowptr, scale, gam, bet, alpha, bias, S2, R2, N, E, W, Hout)
    assert torch.equal(S, S2) and torch.equal(R, R2)
    # reference: per-edge activation -> Linear -> scatter_sum / scatter_mean
    y = xhat.double() * gam.double() + bet.double()
    act = torch.where(y < 0, 0.25 * y, y)
    msg = act @ Wt.double().T + bias.double()
    agg_ref = torch.zeros(N, Hout, device=DEV, dtype=torch.float64).index_add_(0, dst.long(), msg)
    if mean:
        agg_ref = agg_ref * scale.double()[:, None]
    agg = S.double() @ Wt.double().T + R.double()
    assert err(agg, agg_ref) < 2e-5
    # backward: a node-row gradient (a strided column block, like the node MLP's input gradient) expanded per edge
    dcat_n = torch.randn(N, 2 * Hout, device=DEV)
    dagg = dcat_n[:, Hout:]
    daggc = torch.empty(N, Hout, device=DEV)
    ops.seg_count_scale(dagg.data_ptr(), 2 * Hout, rowptr, mean, daggc, N, Hout)
    c = (deg > 0).float() if mean else deg.float()
    assert torch.equal(daggc, dagg * c[:, None])
    dnode = (dagg.double() @ Wt.double()).float()                                        # [N, W]
    rows = ops.ln_prelu_bwd_partial_rows(E)
    pld = 2 * W + 4
    dz, part = torch.empty(E, W, device=DEV), torch.zeros(rows, pld, device=DEV)
    ops.ln_prelu_bwd_gather(dnode, dst, scale, xhat, rstd, gam, bet, alpha, dz, part, E, W)
    dact = dnode[dst.long()] * (scale[dst.long()][:, None] if mean else 1.0)
    dz_ref, part_ref = torch.empty(E, W, device=DEV), torch.zeros(rows, pld, device=DEV)
    ops.ln_prelu_bwd(dact.contiguous(), xhat, rstd, gam, bet, alpha, dz_ref, part_ref, E, W)
    assert torch.equal(dz, dz_ref) and torch.equal(part, part_ref)                        # the same arithmetic on gathered rows


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
    from tests.gpu_util import _fat_crystals
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


@pytest.mark.parametrize("M,H", [(450, 128), (1, 128), (17, 64), (33, 128), (1000, 64), (255, 128), (16, 64)])
@pytest.mark.parametrize("with_res", [True, False])
def test_column_split_node_mlp_forward(M, H, with_res):
    """Round 6: the column-split NodeModel forward - hidden / 16 workgroups per 16-row tile, the pre-LayerNorm
    tile exchanged in-launch (publish / ticket / every sibling waits and reads back) - against the one-workgroup-per-tile kernel
    (fp32 rounding: the k range is split over 4 waves) and float64; with and without the third product (the next layer's node
    products, a second in-launch exchange of the output tile); launched repeatedly: same bits, counters back at zero."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    x, agg, W, r = _node_block(M, H, 11 * M + H)
    res = x if with_res else None
    W1n = r(2 * H, 3 * H) / (3 * H) ** 0.5
    assert ops.mlp_ln_cs(M, 2 * H, 2 * H, H)

    def fwd(cs, third):
        xh, rs, out = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev),
                       torch.full((M, H), float("nan"), device=dev))
        pq = torch.full((M, 4 * H), float("nan"), device=dev) if third else None
        ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], res, xh, rs, out,
                       w3=W1n if third else None, nb3=2 if third else 0, pq=pq, cs=cs)
        torch.cuda.synchronize()
        return out, xh, rs, pq
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref = fwd(False, False)
    for third in (False, True):
        got = fwd(True, third)
        for a_, b_ in zip(got[:3], ref[:3]):
            assert bool(torch.isfinite(a_).all())
            assert float((a_ - b_).abs().max()) <= 5e-6 * sc(b_)
        for _ in range(3):                          # counters are back at zero: the same launch again gives the same bits
            again = fwd(True, third)
            assert all(torch.equal(u, v) for u, v in zip(again[:3], got[:3]))
            assert not third or torch.equal(again[3], got[3])
        if third:
            exact = torch.cat([got[0].double() @ W1n[:, :H].double().t(), got[0].double() @ W1n[:, H:2 * H].double().t()], 1)
            assert float((got[3].double() - exact).abs().max()) <= 2e-5 * sc(exact)
    # float64
    z = torch.cat([x, agg], 1).double() @ W["w1"].double().t() + W["b1"].double()
    y = torch.nn.functional.layer_norm(z, (2 * H,), W["g"].double(), W["b"].double(), 1e-5)
    y = torch.where(y >= 0, y, 0.25 * y)
    o64 = y @ W["w2"].double().t() + W["b2"].double() + (res.double() if res is not None else 0)
    assert float((got[0].double() - o64).abs().max()) <= 2e-5 * sc(o64)


@pytest.mark.parametrize("M,H", [(450, 128), (1, 128), (17, 64), (33, 128), (1000, 64), (255, 128)])
@pytest.mark.parametrize("add_dy", [True, False])
def test_column_split_node_mlp_backward(M, H, add_dy):
    """... and the backward: dz, dcat (+ the residual connection's dy on its first H columns), the summed [dgamma | dbeta | dalpha]
    partial rows against the one-workgroup-per-tile kernel and against float64 autograd; dcat as a strided [M, 2H] view."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    x, agg, W, r = _node_block(M, H, 13 * M + H)
    dy = r(M, H)
    xh, rs, out = torch.empty(M, 2 * H, device=dev), torch.empty(M, device=dev), torch.empty(M, H, device=dev)
    ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows = ops.mlp_ln_bwd_partial_rows(M)
    pld = 4 * H + 4

    def bwd(cs):
        dz = torch.full((M, 2 * H), float("nan"), device=dev)
        dcat = torch.full((M, 2 * H), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        ops.mlp_ln_bwd(M, dy, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=add_dy, cs=cs)
        torch.cuda.synchronize()
        return dz, dcat, part[:, :4 * H].sum(0), part[:, pld - 1].sum()
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref, got = bwd(False), bwd(True)
    for a_, b_ in zip(got, ref):
        assert bool(torch.isfinite(a_).all())
        assert float((a_ - b_).abs().max()) <= 2e-5 * sc(b_)
    again = bwd(True)
    assert all(torch.equal(u, v) for u, v in zip(again, got))
    # float64 autograd
    xa = torch.cat([x, agg], 1).double().requires_grad_(True)
    g64, b64, al64 = (W[k].double().requires_grad_(True) for k in ("g", "b", "al"))
    z = xa @ W["w1"].double().t() + W["b1"].double()
    y = torch.nn.functional.layer_norm(z, (2 * H,), g64, b64, 1e-5)
    o = torch.where(y >= 0, y, al64 * y) @ W["w2"].double().t()
    o.backward(dy.double())
    dcat64 = xa.grad + (torch.cat([dy.double(), torch.zeros(M, H, dtype=torch.float64, device=dev)], 1) if add_dy else 0)
    assert float((got[1].double() - dcat64).abs().max()) <= 2e-5 * sc(dcat64)
    assert float((got[2][:2 * H].double() - g64.grad).abs().max()) <= 5e-5 * sc(g64.grad)
    assert float((got[2][2 * H:].double() - b64.grad).abs().max()) <= 5e-5 * sc(b64.grad)
    assert abs(float(got[3]) - float(al64.grad)) <= 5e-5 * max(1.0, abs(float(al64.grad)))


def test_column_split_exchange_under_a_bandwidth_hog():
    """The in-launch exchange (write-through stores, ticket, agent-scope poll, write-through read-back) 400 times back to back while
    a second stream streams 1 GiB copies through HBM (every XCD's L2 thrashed): every launch bitwise the first."""
    from dostransformer_amd import ops
    dev = "cuda:0"
    M, H = 450, 128
    x, agg, W, r = _node_block(M, H, 77)
    W1n = r(2 * H, 3 * H) / (3 * H) ** 0.5
    dy = r(M, H)
    hog_a, hog_b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
    side = torch.cuda.Stream()
    outs = []
    bad = torch.zeros(1, device=dev)
    first = None
    for it in range(400):
        if it % 8 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        xh, rs, out, pq = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M,), float("nan"), device=dev),
                           torch.full((M, H), float("nan"), device=dev), torch.full((M, 4 * H), float("nan"), device=dev))
        ops.mlp_ln_fwd(M, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], x, xh, rs, out, w3=W1n, nb3=2, pq=pq, cs=True)
        dz, dcat, part = (torch.full((M, 2 * H), float("nan"), device=dev), torch.full((M, 2 * H), float("nan"), device=dev),
                          torch.full((ops.mlp_ln_bwd_partial_rows(M), 4 * H + 4), float("nan"), device=dev))
        ops.mlp_ln_bwd(M, dy, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=True, cs=True)
        cur = (out, xh, pq, dz, dcat, part[:, :4 * H].clone(), part[:, -1].clone())
        if first is None:
            first = cur
        else:
            for u, v in zip(cur, first):
                bad += (u != v).any().float()          # compared on the device: no host sync in the loop
    torch.cuda.synchronize()
    assert float(bad) == 0.0
    assert all(bool(torch.isfinite(t).all()) for t in first)


@pytest.mark.parametrize("n,H,two,add_dy", [(450, 128, False, True), (450, 128, True, False), (37, 64, True, True), (16, 128, False, True),
                                            (3, 64, False, False), (1000, 64, False, True), (261, 128, True, True)])
def test_node_side_gradient_inside_the_column_split_backward_launch(n, H, two, add_dy):
    """DosxMlpLnBwd.pre = 1 (round 6): the node side of the later layer's factored input gradient (dosx_node_grad: source-node
    sums of dz, dx = res + res2 + aggS Wa + aggD Wb) as the front part of the column-split NodeModel backward launch - three
    in-launch exchanges - against the two launches: aggs, dx (= the block's dy), dz, dcat, the summed partial rows; isolated
    nodes, long source segments; twice (same bits, counters back at zero)."""
    import numpy as np
    from dostransformer_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(n + H)
    deg = rng.integers(0, 30, size=n)
    deg[rng.random(n) < 0.15] = 0
    deg[0] = 150 if n > 3 else 5                        # one long source segment (more than 64 edges: two id chunks per wave)
    E = int(deg.sum())
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)).to(dev)
    perm = torch.from_numpy(rng.permutation(E).astype(np.int32)).to(dev)
    x, agg, W, r = _node_block(n, H, 17 * n + H)
    W2 = 2 * H
    dzE, aggd, W0 = r(E, W2), r(n, W2), r(W2, 3 * H) / W2 ** 0.5
    dcat_prev = r(n, W2)
    res2 = r(n, H) if two else None
    xh, rs, out = torch.empty(n, W2, device=dev), torch.empty(n, device=dev), torch.empty(n, H, device=dev)
    ops.mlp_ln_fwd(n, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows, pld = ops.mlp_ln_bwd_partial_rows(n), 4 * H + 4

    def run(fused):
        aggs, dx = torch.full((n, W2), float("nan"), device=dev), torch.full((n, H), float("nan"), device=dev)
        dz, dcat = torch.full((n, W2), float("nan"), device=dev), torch.full((n, W2), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        pre = None
        if fused:
            pre = dict(kind="node_grad", dz=dzE, rowptr_src=rowptr, perm_src=perm, aggd=aggd, w=W0, res=dcat_prev[:, :H], res2=res2, aggs=aggs)
        else:
            ops.node_grad(n, H, dzE, rowptr, perm, aggd, W0, dcat_prev[:, :H], res2, aggs, dx)
        ops.mlp_ln_bwd(n, dx, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=add_dy, cs=True, pre=pre)
        torch.cuda.synchronize()
        return aggs, dx, dz, dcat, part[:, :4 * H].sum(0), part[:, pld - 1].sum()
    sc = lambda t: float(t.abs().max()) + 1e-30
    ref, got = run(False), run(True)
    for name, a_, b_ in zip(("aggs", "dx", "dz", "dcat", "dgamma|dbeta", "dalpha"), got, ref):
        assert bool(torch.isfinite(a_).all()), name
        assert float((a_ - b_).abs().max()) <= 3e-5 * sc(b_), (name, float((a_ - b_).abs().max()) / sc(b_))
    again = run(True)
    assert all(torch.equal(u, v) for u, v in zip(again, got))
    exact = dcat_prev[:, :H].double() + got[0].double() @ W0[:, :H].double() + aggd.double() @ W0[:, H:W2].double()
    if two:
        exact = exact + res2.double()
    assert float((got[1].double() - exact).abs().max()) <= 2e-5 * sc(exact)


@pytest.mark.parametrize("n,H,B", [(450, 128, 64), (37, 64, 5), (16, 128, 1), (1000, 64, 100)])
def test_dense_key_backward_inside_the_column_split_backward_launch(n, H, B):
    """DosxMlpLnBwd.pre = 2 (round 6): dosx_dense_normalize_pool_bwd - the to_dense_batch / key-LayerNorm backward plus the
    pooled decoder gradient, ghost nodes included - as the front part of the last layer's NodeModel backward launch: dx, dz,
    dcat, partial rows bitwise the two launches (row-local: no exchange, the same arithmetic)."""
    import numpy as np
    from dostransformer_amd import ops
    dev = "cuda:0"
    rng = np.random.default_rng(n + H + B)
    x, agg, W, r = _node_block(n, H, 19 * n + H)
    n_ghost = min(5, n // 4)
    n_real = n - n_ghost
    sizes = np.diff(np.sort(np.concatenate([[0, n_real], rng.integers(0, n_real + 1, size=B - 1)])))
    node_graph = np.concatenate([np.repeat(np.arange(B), sizes), np.full(n_ghost, B)]).astype(np.int32)
    nmax = int(sizes.max())
    pos = np.concatenate([np.arange(s) for s in sizes] + [np.zeros(n_ghost, dtype=np.int64)])
    dense_row = np.where(node_graph < B, pos * B + np.minimum(node_graph, B - 1), nmax * B).astype(np.int32)
    dense_row_t, node_graph_t = torch.from_numpy(dense_row).to(dev), torch.from_numpy(node_graph).to(dev)
    dkv, kvhat = r(nmax * B + 1, H), r(nmax * B + 1, H)
    rstd_n = r(n).abs() + 0.5
    Kd = 2 * H
    dpool = r(B, Kd)
    xh, rs, out = torch.empty(n, 2 * H, device=dev), torch.empty(n, device=dev), torch.empty(n, H, device=dev)
    ops.mlp_ln_fwd(n, x, agg, W["w1"], W["b1"], W["g"], W["b"], W["al"], W["w2"], W["b2"], None, xh, rs, out, cs=False)
    rows, pld = ops.mlp_ln_bwd_partial_rows(n), 4 * H + 4

    def run(fused):
        dx = torch.full((n, H), float("nan"), device=dev)
        dz, dcat = torch.full((n, 2 * H), float("nan"), device=dev), torch.full((n, 2 * H), float("nan"), device=dev)
        part = torch.full((rows, pld), float("nan"), device=dev)
        pre = None
        if fused:
            pre = dict(kind="dense", dkv=dkv, kvhat=kvhat, rstd_nodes=rstd_n, dense_row=dense_row_t, dpool_ptr=dpool.data_ptr() + 4 * (Kd - H),
                       ld_dpool=Kd, node_graph=node_graph_t, num_graphs=B, ghost_row=nmax * B)
        else:
            ops.dense_normalize_pool_bwd(dkv, kvhat, rstd_n, dense_row_t, dpool.data_ptr() + 4 * (Kd - H), Kd, node_graph_t, B, dx, n, H,
                                         False, ghost_row=nmax * B)
        ops.mlp_ln_bwd(n, dx, xh, rs, W["w1"], W["w2"], W["g"], W["b"], W["al"], dz, dcat, part, add_dy=True, cs=True, pre=pre)
        torch.cuda.synchronize()
        return dx, dz, dcat, part[:, :4 * H].clone(), part[:, pld - 1].clone()
    ref, got = run(False), run(True)
    sc = lambda t: float(t.abs().max()) + 1e-30
    for name, a_, b_ in zip(("dx", "dz", "dcat", "partials", "dalpha"), got, ref):
        assert bool(torch.isfinite(a_).all()), name
        assert float((a_ - b_).abs().max()) <= 2e-6 * sc(b_), name          # (row sums over 16 instead of 64 lanes: rounding)
    assert float(got[0][n_real:].abs().max()) == 0.0 if n_ghost else True    # ghost nodes: zero gradient


@pytest.mark.parametrize("M,Fa,H", [(450, 118, 128), (1, 118, 128), (37, 118, 64), (255, 200, 128), (1000, 64, 64), (16, 6, 128)])
def test_node_encoder_and_first_node_products_in_one_column_split_launch(M, Fa, H):
    """dosx_enc_cs_fwd (round 6): Linear(Fa, H) -> PReLU -> Linear(H, H) of the node encoder (DOSTransformer_phonon.py:129,141) and
    layer 0's node products x0 [Wa | Wb]^T as ONE column-split launch with two in-launch exchanges, against the two dosx_gemm +
    dosx_gemm_pair launches and float64; Fa = 118 (rows of 472 bytes: 8-byte fragment loads, zeroed tail); twice (same bits)."""
    from dostransformer_amd import functional as Fn
    o = ops()
    x = rnd(M, Fa, seed=1)
    P = {"k.0.weight": rnd(H, Fa, seed=2, scale=Fa ** -0.5), "k.0.bias": rnd(H, seed=3, scale=0.1), "k.1.weight": torch.tensor([0.25], device=DEV),
         "k.2.weight": rnd(H, H, seed=4, scale=H ** -0.5), "k.2.bias": rnd(H, seed=5, scale=0.1)}
    W1 = rnd(2 * H, 3 * H, seed=6, scale=(3 * H) ** -0.5)
    assert o.enc_cs_supported(M, Fa, H)
    y0, ctx0 = Fn.mlp_prelu_fwd(P, "k", Fn.SegList([o.seg(x)], [x]), M, H)
    pq0 = torch.empty(M, 4 * H, device=DEV)
    o.gemm_pair(dict(M=M, N=2 * H, segs=[o.seg(y0)], w=W1[:, :H], out=pq0[:, :2 * H]),
                dict(M=M, N=2 * H, segs=[o.seg(y0)], w=W1[:, H:2 * H], out=pq0[:, 2 * H:]))
    got = []
    for _ in range(3):
        z, y, pq = (torch.full((M, H), float("nan"), device=DEV), torch.full((M, H), float("nan"), device=DEV),
                    torch.full((M, 4 * H), float("nan"), device=DEV))
        o.enc_cs_fwd(M, x, P["k.0.weight"], P["k.0.bias"], P["k.1.weight"], P["k.2.weight"], P["k.2.bias"], z, y, W1, pq)
        torch.cuda.synchronize()
        got.append((z, y, pq))
    z, y, pq = got[0]
    assert all(torch.equal(u, v) for g_ in got[1:] for u, v in zip(g_, got[0]))
    assert bool(torch.isfinite(z).all()) and bool(torch.isfinite(y).all()) and bool(torch.isfinite(pq).all())
    assert err(z, ctx0[1]) < 5e-6 and err(y, y0) < 5e-6 and err(pq, pq0) < 1e-5
    z64 = x.double() @ P["k.0.weight"].double().T + P["k.0.bias"].double()
    y64 = torch.where(z64 >= 0, z64, 0.25 * z64) @ P["k.2.weight"].double().T + P["k.2.bias"].double()
    pq64 = torch.cat([y64 @ W1[:, :H].double().T, y64 @ W1[:, H:2 * H].double().T], 1)
    assert err(z, z64) < 2e-5 and err(y, y64) < 2e-5 and err(pq, pq64) < 2e-5


@pytest.mark.parametrize("E,H", [(9000, 128), (1, 128), (37, 64), (1281, 64)])
def test_phonon_edge_encoder_in_one_launch(E, H):
    """dosx_edge_enc_fwd (round 6): SH(l <= 1) * smooth_cutoff features, the K = 4 Linear, PReLU and the second Linear of the phonon edge
    encoder (DOSTransformer_phonon.py:74-77,129,142) as ONE launch == dosx_edge_embed_sh1 + dosx_gemm (PReLU prologue): the features
    and the pre-activation bitwise (the same fma chains), the output to rounding; zero-length self edges and all cutoff regimes."""
    from dostransformer_amd import functional as Fn
    o = ops()
    vec = rnd(E, 3, seed=1, scale=4.0 / 3 ** 0.5)
    vec[::7] = 0.0                                      # self-interaction edges: zero vectors
    P = {"k.0.weight": rnd(H, 4, seed=2, scale=0.5), "k.0.bias": rnd(H, seed=3, scale=0.1), "k.1.weight": torch.tensor([0.25], device=DEV),
         "k.2.weight": rnd(H, H, seed=4, scale=H ** -0.5), "k.2.bias": rnd(H, seed=5, scale=0.1)}
    ea0, z0 = o.edge_embed_sh1(vec, P["k.0.weight"], P["k.0.bias"], 4.0)
    e0, _ = Fn.mlp_prelu_fwd(P, "k", Fn.SegList([o.seg(ea0)], [ea0]), E, H, z=z0)
    ea1, z1, e1 = o.edge_enc_fwd(vec, P["k.0.weight"], P["k.0.bias"], P["k.1.weight"], P["k.2.weight"], P["k.2.bias"], 4.0)
    torch.cuda.synchronize()
    assert torch.equal(ea1, ea0) and torch.equal(z1, z0)
    assert bool(torch.isfinite(e1).all()) and err(e1, e0) < 5e-6
    z64 = z0.double()
    ref = torch.where(z64 >= 0, z64, 0.25 * z64) @ P["k.2.weight"].double().T + P["k.2.bias"].double()
    assert err(e1, ref) < 2e-5
