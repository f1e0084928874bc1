"""CPU tests of the host-side logic: batch container / CSR metadata, drop-in module structure,
flat-parameter storage, data-parallel sharding."""
import numpy as np
import pytest
import torch

from dostransformer_amd import synth
from dostransformer_amd.batch import CrystalBatch, collate, graph_meta, split_crystals
from dostransformer_amd.dist import shard_batch, shard_bounds


def test_batch_mapping_and_attribute_protocol():
    g = synth.phonon_batch(4, seed=1, dtype=torch.float64)
    assert "batch" in g and "edge_index" in g and "pos" not in g          # DOSTransformer_phonon.py:48-56
    assert g["edge_vec"] is g.edge_vec and g.x.shape[1] == 118 and g.phdos.shape == (4, 51)
    assert g.system.shape == (4,) and g.system.dtype == torch.int64
    e = synth.edos_batch(3, seed=2)
    assert e.glob.shape == (6,) and e.y_ft.shape == (3 * 201,) and len(e.mp_id) == 3 and e.x.shape[1] == 200
    counts = torch.bincount(e.batch)
    ptr = torch.cumsum(counts, 0)
    assert torch.all(e.x[ptr - 1] == 0)                                    # phantom all-zero node per crystal
    assert not torch.isin(ptr - 1, e.edge_index.flatten()).any()           # ... and it is isolated
    g.to("cpu")


@pytest.mark.parametrize("sort_edges", [True, False])
def test_csr_metadata(sort_edges):
    g = synth.phonon_batch(5, seed=3, dtype=torch.float32, sort_edges=sort_edges)
    m = graph_meta(g)
    N, E, B = m.num_nodes, m.num_edges, m.num_graphs
    src, dst = m.src.long(), m.dst.long()
    assert torch.all(dst[1:] >= dst[:-1])
    ei = g.edge_index if m.edge_perm is None else g.edge_index[:, m.edge_perm]
    assert torch.equal(ei[0], src) and torch.equal(ei[1], dst)
    rp = m.rowptr_dst.long()
    for n in range(N):
        assert torch.all(dst[rp[n]:rp[n + 1]] == n)
    assert rp[-1] == E
    rs = m.rowptr_src.long()
    ps = m.perm_src.long()
    for n in range(N):
        assert torch.all(src[ps[rs[n]:rs[n + 1]]] == n)
    assert sorted(ps.tolist()) == list(range(E))
    gp = m.graph_ptr.long()
    assert gp[-1] == N and m.n_max == int((gp[1:] - gp[:-1]).max())
    pos = torch.arange(N) - gp[m.node_graph.long()]
    assert torch.equal(m.dense_row.long(), pos * B + m.node_graph.long())
    deg = torch.bincount(dst, minlength=N).clamp(min=1).float()
    assert torch.allclose(m.inv_deg, 1.0 / deg)
    with pytest.raises(ValueError):
        collate(split_crystals(g), n_max=1)


def test_foreign_batch_object_gets_metadata():
    g = synth.edos_batch(3, seed=4, sort_edges=False)

    class Foreign:            # e.g. a PyG Batch: attributes only
        pass
    f = Foreign()
    for k in g.keys():
        setattr(f, k, g[k])
    m = graph_meta(f)
    assert m.num_graphs == 3 and m.edge_perm is not None and m.num_edges == g.edge_index.shape[1]


def test_split_collate_roundtrip():
    g = synth.edos_batch(4, seed=5)
    g2 = collate(split_crystals(g))
    for k in ("x", "edge_index", "edge_attr", "glob", "y_ft", "system", "batch"):
        assert torch.equal(g[k], g2[k]), k


def test_shard_bounds_cover_and_balance():
    for n, w in [(64, 8), (8, 8), (512, 8), (10, 3), (5, 2)]:
        ne = np.random.RandomState(n).randint(40, 240, size=n).tolist()
        b = shard_bounds(ne, w)
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert all(hi > lo for lo, hi in b)
        loads = [sum(ne[lo:hi]) for lo, hi in b]
        if n >= 8 * w:
            assert max(loads) < 1.35 * sum(ne) / w
    cs = synth.phonon_crystals(16, seed=6, dtype=torch.float32)
    n_max = max(c["x"].shape[0] for c in cs)
    shards = [shard_batch(cs, 4, r) for r in range(4)]
    assert sum(s.num_graphs for s in shards) == 16 and all(s.meta.n_max == n_max for s in shards)


def test_flat_params_layout_and_dead_parameters():
    from dostransformer_amd._fused import FlatParams, is_dead_param
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    model = DOSTransformer_phonon(3, 2, 118, 4, 16, "cpu", 0.0)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    fp = FlatParams(model, torch.device("cpu"))
    after = model.state_dict()
    assert list(before) == list(after) and all(torch.equal(before[k], after[k]) for k in before)
    dead = [n for n, _ in model.named_parameters() if is_dead_param(n)]
    n_dead = sum(p.numel() for n, p in model.named_parameters() if is_dead_param(n))
    assert "alpha" in dead and any("node_mlp_1" in d for d in dead) and any("in_proj_weight" in d for d in dead)
    assert all(n not in fp.P for n in dead) and n_dead > 0
    for n, p in model.named_parameters():
        if n in fp.P:
            assert p.data_ptr() == fp.P[n].data_ptr() and fp.G[n].shape == p.shape
            assert (p.data_ptr() - fp.flat.data_ptr()) % 256 == 0
    # in-place updates of the flat buffer are what the module sees (fused AdamW writes there)
    fp.flat.add_(1.0)
    assert torch.allclose(model.fc.weight, before["fc.weight"] + 1.0)
    assert fp.intact(torch.device("cpu"))
    model.double()
    assert not fp.intact(torch.device("cpu"))            # .to()/.double() re-homes parameters -> re-flatten


def test_drop_in_aliases():
    import sys
    import dostransformer_amd
    dostransformer_amd.install_dropin()
    from layers import TransformerEncoder                                   # reference: DOSTransformer.py:6
    from embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon  # reference: main_phDOS.py:67
    from embedder_eDOS.DOSTransformer import DOSTransformer                 # reference: main_eDOS.py:68
    from embedder_eDOS.graphnetwork import Graphnetwork
    from embedder_phDOS.graphnetwork_phonon import Graphnetwork_phonon
    assert TransformerEncoder.__module__.startswith("dostransformer_amd")
    for k in ("layers", "embedder_phDOS", "embedder_eDOS"):
        sys.modules.pop(k, None)


def test_flat_params_bucket_layout():
    """[last = encoders + layer 0 | mid = rest of the GNN trunk | early = the rest], all aligned, every live parameter exactly once."""
    import torch
    from dostransformer_amd._fused import FlatParams, is_dead_param, is_last_param, is_late_param
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    model = DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0)
    fp = FlatParams(model, torch.device("cpu"))
    live = [n for n, _ in model.named_parameters() if not is_dead_param(n)]
    assert sorted(fp.names) == sorted(live)
    flags = [is_late_param(n) for n in fp.names]
    assert flags == sorted(flags, reverse=True) and any(flags) and not all(flags)      # late block first
    for n, o in zip(fp.names, fp.offsets):
        assert (o < fp.n_late) == is_late_param(n) and o % 64 == 0
    assert fp.n_late % 64 == 0 and 0 < fp.n_late < fp.total
    # the GNN trunk's slice again in two: what is final only at the end of the backward pass (encoders, layer 0) comes first
    for n, o in zip(fp.names, fp.offsets):
        assert (o < fp.n_last) == is_last_param(n)
    assert fp.n_last % 64 == 0 and 0 < fp.n_last < fp.n_late
    mid = [n for n, o in zip(fp.names, fp.offsets) if fp.n_last <= o < fp.n_late]
    assert any(n.startswith("GN_decoder.") for n in mid) and any(n.startswith("stacked_processor.2.") for n in mid)
    assert not any(n.startswith(("GN_encoder.", "stacked_processor.0.")) for n in mid)
    # at the headline shape (hidden 128, 3 layers) the exposed bucket is 1.12 MB of the trunk's 3.04 MB
    fp2 = FlatParams(DOSTransformer_phonon(3, 2, 118, 4, 128, "cpu", 0.0), torch.device("cpu"))
    assert 4 * fp2.n_last == 1121280 and 4 * (fp2.n_late - fp2.n_last) == 1916416 and 4 * (fp2.total - fp2.n_late) == 3501056


def test_checkpoint_is_read_without_unpickling_objects(tmp_path):
    """checkpoint.load uses weights_only=True: a file carrying an arbitrary pickled object is refused unless
    the caller opts in; our own format (tensors / str / int / float / tuple / dict) round-trips."""
    from dostransformer_amd import checkpoint
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    torch.manual_seed(0)
    m = DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0)
    p = str(tmp_path / "a.pt")
    checkpoint.save(p, m, extra={"epoch": 3, "note": "x"})
    m2 = DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0)
    assert checkpoint.load(p, m2) == {"epoch": 3, "note": "x"}
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k

    class Evil:
        def __reduce__(self):
            return (print, ("pwned",))
    bad = str(tmp_path / "bad.pt")
    torch.save({"model": m.state_dict(), "extra": {"x": Evil()}}, bad)
    with pytest.raises(Exception):
        checkpoint.load(bad, m2)


def test_padded_copy_cache_is_invalidated_by_assignment():
    g = synth.phonon_batch(2, seed=5, dtype=torch.float32)
    object.__setattr__(g, "_dosx_padded", ("bucket", "copy"))
    g.to("cpu")
    assert g._dosx_padded is not None                      # nothing moved
    g.x = g.x * 2
    assert g._dosx_padded is None
    object.__setattr__(g, "_dosx_padded", ("bucket", "copy"))
    g["edge_vec"] = g.edge_vec + 1
    assert g._dosx_padded is None
    object.__setattr__(g, "_dosx_padded", ("bucket", "copy"))
    g.to("cpu", torch.float64)
    assert g._dosx_padded is None and g.x.dtype == torch.float64


def test_shard_batch_records_the_global_batch_size():
    from dostransformer_amd.batch import bucket_sizes, pad_batch
    cs = synth.phonon_crystals(7, seed=4, dtype=torch.float32)
    g = shard_batch(cs, 2, 1)
    assert g.n_global == 7
    gp = pad_batch(g, *bucket_sizes(g.meta.num_nodes, g.meta.num_edges))
    assert gp.n_global == 7 and gp.clone().n_global == 7
    assert collate(cs).n_global is None


def test_optimizer_moments_follow_rehomed_parameters():
    """Trainer._state carries m / v BY NAME when the flat parameter buffer is rebuilt (CPU-only: pure bookkeeping)."""
    from dostransformer_amd._fused import FlatParams
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.train import Trainer
    torch.manual_seed(0)
    m = DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0)
    tr = Trainer(m)
    fp = m.flat_params()
    mm, vv = tr._state(fp)
    mm.copy_(torch.arange(mm.numel(), dtype=torch.float32))
    vv.fill_(2.0)
    tr.step_count = 5
    want = {n: mm[o:o + fp.P[n].numel()].clone() for n, o in zip(fp.names, fp.offsets)}
    fp2 = FlatParams(m, torch.device("cpu"), extra_dead=("fc.bias",))       # a different layout (one parameter less)
    m2, v2 = tr._state(fp2)
    assert tr.step_count == 5 and m2.numel() == fp2.total
    for n, o in zip(fp2.names, fp2.offsets):
        assert torch.equal(m2[o:o + fp2.P[n].numel()], want[n]), n
        assert float(v2[o:o + fp2.P[n].numel()].min()) == 2.0


def test_seg_tiles_host_properties():
    """Node-aligned row tiles of the message GEMM (batch.seg_tiles_host), incl. over-full nodes (> 48 incoming edges) cut into
    chunk tiles and isolated nodes: monotone boundaries, <= 48 rows per tile, every node owned exactly once, every edge covered
    exactly once, chunk k of a node starts 48*k rows into its segment whatever surrounds it, consecutive tiles hold more than
    48 rows together (what seg_tile_bound relies on)."""
    from dostransformer_amd.batch import SEG_TILE_ROWS, seg_tile_bound, seg_tiles_host
    R = SEG_TILE_ROWS
    rng = np.random.default_rng(0)
    for trial in range(400):
        n = int(rng.integers(1, 30))
        deg = rng.integers(0, 30, size=n)
        deg[rng.random(n) < 0.3] = 0
        for _ in range(int(rng.integers(0, 5))):
            deg[int(rng.integers(0, n))] = int(rng.choice([48, 49, 60, 96, 97, 144, 145, 200]))
        rp = np.concatenate([[0], np.cumsum(deg)])
        t = seg_tiles_host(rp)
        assert t.shape[0] == 3 and t.dtype == np.int32
        eb, nb, pi = t
        T = t.shape[1] - 1
        assert eb[0] == 0 and nb[0] == 0 and eb[-1] == rp[-1] and nb[-1] == n and pi[-1] == 0
        assert (np.diff(eb) >= 0).all() and (np.diff(nb) >= 0).all() and np.diff(eb).max(initial=0) <= R
        cover, owned = np.zeros(n, int), np.zeros(n, int)
        for i in range(T):
            lo, hi = nb[i], nb[i + 1]
            whole = range(lo, hi)
            if pi[i]:
                ci, nc = pi[i] >> 16, pi[i] & 0xffff
                assert deg[lo] > R and nc == -(-deg[lo] // R) and 0 <= ci < nc
                assert eb[i] == rp[lo] + ci * R                        # the chunk is cut from the START of the node's segment
                rows = min(rp[lo + 1], eb[i + 1]) - eb[i]
                assert rows == (R if ci < nc - 1 else deg[lo] - ci * R)
                assert (hi == lo) if ci < nc - 1 else (hi >= lo + 1)
                cover[lo] += rows
                owned[lo] += ci == nc - 1
                whole = range(lo + 1, hi)
            for k in whole:
                assert deg[k] <= R and rp[k] >= eb[i] and rp[k + 1] <= eb[i + 1]
                cover[k] += deg[k]
                owned[k] += 1
        assert (cover == deg).all() and (owned == 1).all()
        rows = np.diff(eb)
        assert all(i == 0 and rows[0] == 0 for i in range(T - 1) if rows[i] + rows[i + 1] <= R)
        assert T <= seg_tile_bound(n, int(rp[-1]), 1)


def test_size_policies_of_the_factored_forms():
    """The size limits that switch the EdgeModel between its fused and its factored forms, and the tail split of the unfused
    feed-forward layers (functional._factor_edge / _factor_last / _ffn_tail_start): the BASELINE shapes land where DESIGN.md
    says they do."""
    from dostransformer_amd import functional as Fn
    # Phonon-DOS benchmark shape (H 128, ~9000 edges): factored first Linear since round 5 (gathers / node sums inside the GEMM
    # epilogues), per-edge last layer; the CPU-reference shape (H 64, ~1100 edges: 0.05 GF) keeps the gathered-concat GEMM;
    # Electron-DOS (H 256, 17880 edges; the 32-crystal shard has about half of them): factored first Linear and aggregate-first
    # last layer
    assert Fn._factor_edge(9000, 128) and not Fn._factor_last(9000, 128)
    assert not Fn._factor_edge(1100, 64) and not Fn._factor_last(1100, 64)

    class _M:                                                  # a batch with node-aligned row tiles: the one-launch kernels of
        seg_tile = object()                                    # csrc/edge_mlp.hip take hidden 64 / 128 at EVERY size
    assert Fn._factor_edge(1100, 64, _M()) and Fn._factor_edge(140, 128, _M())
    assert not Fn._factor_edge(300, 256, _M()) and Fn._factor_edge(1100, 256, _M())    # hidden 256: the epilogue forms, by size (0.5 GF)
    assert Fn._factor_edge(17880, 256) and Fn._factor_last(17880, 256)
    assert Fn._factor_edge(8900, 256) and Fn._factor_last(8900, 256)
    assert not Fn._factor_last(17880, 384)                    # 2 * hidden > 512: the unfused wide-row path keeps the per-edge form
    # output heads: factored at the Electron-DOS shapes (B * S = 64 * 201 and 32 * 201 rows, H 256), not at 64 * 51 rows / H 128
    assert Fn._factor_heads(64 * 201, 256) and Fn._factor_heads(32 * 201, 256)
    assert not Fn._factor_heads(64 * 51, 128) and not Fn._factor_heads(51, 128)
    # feed-forward tail: 25728 = 3 full rounds of 8192 rows + 1152; 12864 = 1 round + 4672 (too large a tail); small problems never
    assert Fn._ffn_tail_start(25728, 256) == 24576
    assert Fn._ffn_tail_start(12864, 256) == 0
    assert Fn._ffn_tail_start(8192, 256) == 0 and Fn._ffn_tail_start(6528, 256) == 0
    assert Fn._ffn_tail_start(25728, 384) == 0                # 3 column tiles do not divide the 256 CUs: no split
    assert Fn._ffn_tail_start(16384 + 100, 512) == 16384      # 4 column tiles: rounds of 4096 rows
    # round 5: the NodeModel FORWARD runs as one launch up to hidden 256 (1554 nodes of the Electron-DOS batch), its backward only up
    # to hidden 128; unfused encoder layers (hidden 256) take LN1 from the attention kernel; the measured-neutral forms are off
    from dostransformer_amd import ops
    assert ops.mlp_ln_fwd_supported(1554, 512, 512, 256) and not ops.mlp_ln_supported(1554, 512, 512, 256)
    assert ops.mlp_ln_supported(450, 256, 256, 128) and not ops.mlp_ln_fwd_supported(5000, 256, 256, 128)
    assert Fn._LN1_IN_ATTN and not Fn._LN1_WGRAD and not Fn._PQ_IN_NODE_MLP and Fn._ENC_BWD_PAIR


def test_bucket_promotion_picks_the_smallest_live_bucket_that_fits():
    """train.promote_key: a first-time bucket runs in a live one of the same batch size / key-slot count / tiling with at
    least as many rows and at most `tol` more of either, the smallest by (edges, nodes); never in a smaller or a foreign one."""
    from dostransformer_amd.train import promote_key
    rest = (64, 12, 64, True)
    live = [(448, 8960) + rest, (464, 9280) + rest, (480, 9600) + rest, (464, 9280, 32, 12, 32, True), (432, 8640) + rest]
    assert promote_key(live, (448, 9280) + rest, 0.08) == (464, 9280) + rest          # exact edges, one node step up
    assert promote_key(live, (448, 9000) + rest, 0.08) == (464, 9280) + rest          # (448, 8960) has too few edges
    assert promote_key(live, (432, 8640) + rest, 0.08) == (432, 8640) + rest          # itself, if it were live
    assert promote_key(live, (480, 9920) + rest, 0.08) is None                        # nothing large enough
    assert promote_key(live, (400, 8000) + rest, 0.08) == (432, 8640) + rest          # exactly 8 % more of both
    assert promote_key(live, (400, 7900) + rest, 0.08) is None                        # ... and just beyond
    assert promote_key(live, (464, 9280, 16, 12, 16, True), 0.5) is None              # other batch size: never
    assert promote_key([], (448, 8960) + rest, 0.08) is None
