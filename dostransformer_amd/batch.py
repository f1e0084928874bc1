"""Crystal-graph batch container, collate and CSR metadata.

This is the build's counterpart of the PyG ``Batch`` object that the reference's
models receive (`main_eDOS.py:54,104-109`, `main_phDOS.py:53,104-107`): a bag of
concatenated per-crystal tensors that exposes BOTH attribute access (``g.x``,
``g.batch``, ``g.system`` — `DOSTransformer_phonon.py:79,86,105`) and mapping
access (``'batch' in data``, ``data['edge_index']`` —
`DOSTransformer_phonon.py:48-56`).

On top of the reference schema it carries the graph metadata the HIP kernels
want, computed once at collate time on the host so that the forward pass has no
host synchronisation (the reference syncs on ``batch.unique()``
`DOSTransformer_phonon.py:143` and inside ``to_dense_batch`` `:86`):

* edges physically sorted by destination (``col = edge_index[1]``) so the
  edge->node aggregation (`DOSTransformer_phonon.py:209`, `DOSTransformer.py:187`)
  is a contiguous CSR segment reduction (no atomics, bitwise reproducible);
* ``rowptr_dst [N+1]``, and the source-sorted inverse index
  ``perm_src [E]`` / ``rowptr_src [N+1]`` for the gather-backward scatter-add;
* ``graph_ptr [B+1]``, per-node graph id / position, ``n_max`` (global max atoms
  per crystal — it changes the numerics because the reference attends over the
  zero-padded rows).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

_META_KEY = "_dosx_meta"


@dataclass
class GraphMeta:
    """Index metadata for one batch, int32, resident on the batch's device."""
    num_nodes: int
    num_edges: int
    num_graphs: int
    n_max: int
    src: torch.Tensor          # [E] int32, edge source  (row = edge_index[0]) in dst-sorted order
    dst: torch.Tensor          # [E] int32, edge dest    (col = edge_index[1]) non-decreasing
    edge_perm: Optional[torch.Tensor]  # [E] int64: dst-sorted position -> position in the caller's edge arrays (None = identity)
    rowptr_dst: torch.Tensor   # [N+1] int32
    perm_src: torch.Tensor     # [E] int32: ids (in dst-sorted numbering) ordered by src
    rowptr_src: torch.Tensor   # [N+1] int32
    graph_ptr: torch.Tensor    # [B+1] int32
    node_graph: torch.Tensor   # [N] int32
    dense_row: torch.Tensor    # [N] int32: row of node n in the [Nmax*B] dense layout = pos*B + graph
    inv_deg: torch.Tensor      # [N] float32: 1/max(in-degree,1)  (scatter_mean divisor)
    # [3, T+1] int32 or None: node-aligned row tiles of the message GEMM (DosxGemm EPI_SEGSUM): tile t owns the edges
    # [seg_tile[0,t], seg_tile[0,t+1]) (<= SEG_TILE_ROWS) = the whole destination segments of the nodes
    # [seg_tile[1,t], seg_tile[1,t+1]); seg_tile[2,t] != 0 (= chunk << 16 | chunks): the tile's FIRST node, seg_tile[1,t],
    # has more incoming edges than a tile holds and this tile owns chunk `chunk` of its `chunks` row chunks (all but the last
    # are full tiles with no other node; the last one goes on with whole nodes) - see seg_tiles_host.  None: the metadata
    # came from a builder that does not produce tiles (the layers then run the stand-alone segment reduction).
    seg_tile: Optional[torch.Tensor] = None

    def to(self, device) -> "GraphMeta":
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return GraphMeta(**kw)


SEG_TILE_ROWS = 48            # rows of a node-aligned tile = BMR of gemm_kernel<3, ...> (three 16-row MFMA sub-tiles)


def seg_tile_bound(n_pad: int, e_pad: int, num_graphs: int) -> int:
    """Number of tile slots T of a shape bucket (the message GEMM's grid): any two consecutive greedy tiles of a crystal hold
    more than SEG_TILE_ROWS rows together (except a leading tile of isolated nodes in front of an over-full one),
    crystal-aligned tilings add at most one tile per crystal, + the ghost / node-only tiles."""
    return (2 * e_pad + SEG_TILE_ROWS - 1) // SEG_TILE_ROWS + 2 * num_graphs + 2


def seg_tiles_host(rowptr: np.ndarray, rows: int = SEG_TILE_ROWS) -> np.ndarray:
    """Greedy node-aligned tiling of destination-sorted edges: [3, T+1] = edge boundaries, node boundaries, chunk info.

    A node with MORE than ``rows`` incoming edges (periodic neighbour lists at r_max = 4 A produce them, `utils.py:267`)
    closes the running tile and is cut into chunks of ``rows`` edges from the start of its segment: every full chunk is a
    tile of its own that lists no whole node (node boundaries k, k), the remainder (if any) opens a tile that goes on
    greedily with the following whole nodes.  Row 2 marks those tiles: ``chunk << 16 | chunks`` says that the tile's first
    node is that node and which of its chunks the tile holds; the message GEMM then adds the chunk sums of a node in chunk
    order (csrc/gemm.hip, EPI_SEGSUM), so a node's aggregate does not depend on what else shares the batch.  Any two
    consecutive tiles still hold more than ``rows`` rows together (seg_tile_bound)."""
    n = int(rowptr.shape[0]) - 1
    eb, nb, pi, k = [0], [0], [], 0
    while k < n:
        deg = int(rowptr[k + 1] - rowptr[k])
        if deg > rows:
            nfull, r = divmod(deg, rows)
            nc = nfull + (1 if r else 0)
            if nc >= (1 << 16):
                raise ValueError(f"node {k} has {deg} incoming edges: more than {rows} * 65535")
            for i in range(nfull):
                pi.append((i << 16) | nc)
                eb.append(int(rowptr[k]) + (i + 1) * rows)
                nb.append(k + 1 if i == nc - 1 else k)
            if not r:
                k += 1
                continue
            lim = int(rowptr[k]) + nfull * rows + rows      # the remainder's tile goes on with whole nodes
            info = (nfull << 16) | nc
        else:
            lim = int(rowptr[k]) + rows
            info = 0
        j = int(np.searchsorted(rowptr, lim, side="right")) - 1          # largest j with rowptr[j] <= lim
        j = min(max(j, k + 1), n)
        if info == 0 and int(rowptr[j]) == eb[-1] and pi and pi[-1] != 0:
            # only isolated nodes (no rows) in front of the next over-full node, right behind the full last chunk of the
            # previous one: that tile takes them as whole nodes (a tile without rows would break seg_tile_bound's pairing)
            nb[-1] = j
            k = j
            continue
        pi.append(info)
        eb.append(int(rowptr[j]))
        nb.append(j)
        k = j
    return np.stack([np.asarray(eb, np.int32), np.asarray(nb, np.int32), np.asarray(pi + [0], np.int32)])


def _build_meta_host(edge_index: np.ndarray, batch: np.ndarray, num_graphs: int,
                     n_max: Optional[int], presorted: bool) -> GraphMeta:
    n = int(batch.shape[0])
    e = int(edge_index.shape[1])
    src = edge_index[0].astype(np.int64)
    dst = edge_index[1].astype(np.int64)
    if presorted:
        perm = None
    else:
        perm = np.argsort(dst, kind="stable")
        src, dst = src[perm], dst[perm]
    deg_in = np.bincount(dst, minlength=n) if e else np.zeros(n, np.int64)
    rowptr_dst = np.zeros(n + 1, np.int64)
    np.cumsum(deg_in, out=rowptr_dst[1:])
    perm_src = np.argsort(src, kind="stable")
    deg_out = np.bincount(src, minlength=n) if e else np.zeros(n, np.int64)
    rowptr_src = np.zeros(n + 1, np.int64)
    np.cumsum(deg_out, out=rowptr_src[1:])
    counts = np.bincount(batch, minlength=num_graphs)
    graph_ptr = np.zeros(num_graphs + 1, np.int64)
    np.cumsum(counts, out=graph_ptr[1:])
    true_max = int(counts.max()) if num_graphs else 0
    if n_max is None:
        n_max = true_max
    elif n_max < true_max:
        raise ValueError(f"n_max={n_max} smaller than the largest crystal ({true_max} atoms)")
    pos = np.arange(n, dtype=np.int64) - graph_ptr[batch]
    dense_row = pos * num_graphs + batch
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32)))
    return GraphMeta(
        num_nodes=n, num_edges=e, num_graphs=num_graphs, n_max=int(n_max),
        src=i32(src), dst=i32(dst),
        edge_perm=None if perm is None else torch.from_numpy(perm.astype(np.int64)),
        rowptr_dst=i32(rowptr_dst), perm_src=i32(perm_src), rowptr_src=i32(rowptr_src),
        graph_ptr=i32(graph_ptr), node_graph=i32(batch), dense_row=i32(dense_row),
        inv_deg=torch.from_numpy((1.0 / np.maximum(deg_in, 1)).astype(np.float32)),
        seg_tile=torch.from_numpy(seg_tiles_host(rowptr_dst)),
    )


class CrystalBatch:
    """Attribute + mapping bag of batched crystal-graph tensors (see module doc)."""

    def __init__(self, fields: Dict[str, object], num_graphs: int, meta: Optional[GraphMeta] = None):
        object.__setattr__(self, "_fields", dict(fields))
        object.__setattr__(self, "num_graphs", int(num_graphs))
        object.__setattr__(self, _META_KEY, meta)
        object.__setattr__(self, "n_global", None)         # crystals in the un-sharded batch (set by dist.shard_batch)
        object.__setattr__(self, "_dosx_padded", None)     # predict.Predictor's cached ghost-padded copy

    # --- mapping protocol (`'batch' in data`, `data['edge_index']`) ---
    def __contains__(self, key):
        return key in self._fields

    def __getitem__(self, key):
        return self._fields[key]

    def __setitem__(self, key, value):
        self._fields[key] = value
        object.__setattr__(self, "_dosx_padded", None)     # (contents changed: the cached padded copy is stale)

    def keys(self):
        return self._fields.keys()

    # --- attribute protocol (`g.x`, `g.batch`) ---
    def __getattr__(self, name):
        f = object.__getattribute__(self, "_fields")
        if name in f:
            return f[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in ("num_graphs", _META_KEY, "n_global"):
            object.__setattr__(self, name, value)
        else:
            self._fields[name] = value
        object.__setattr__(self, "_dosx_padded", None)     # (contents / metadata changed: cached padded copy is stale)

    @property
    def meta(self) -> Optional[GraphMeta]:
        return object.__getattribute__(self, _META_KEY)

    def to(self, device, dtype: Optional[torch.dtype] = None) -> "CrystalBatch":
        """In-place move like PyG's ``batch.to(device)`` (`main_eDOS.py:106`); returns self."""
        moved = False
        for k, v in list(self._fields.items()):
            if isinstance(v, torch.Tensor):
                w = v
                if dtype is not None and w.is_floating_point():
                    w = w.to(dtype)
                w = w.to(device)
                if w is not v:
                    self._fields[k] = w
                    moved = True
        m = self.meta
        if m is not None and m.src.device != torch.empty(0, device=device).device:
            object.__setattr__(self, _META_KEY, m.to(device))
            moved = True
        if moved:           # (evaluation loops call .to(device) on every visit: keep Predictor's padded copy when nothing moved)
            object.__setattr__(self, "_dosx_padded", None)
        return self

    def clone(self) -> "CrystalBatch":
        f = {k: (v.clone() if isinstance(v, torch.Tensor) else list(v) if isinstance(v, list) else v)
             for k, v in self._fields.items()}
        out = CrystalBatch(f, self.num_graphs, self.meta)
        object.__setattr__(out, "n_global", self.n_global)
        return out

    def __repr__(self):
        parts = []
        for k, v in self._fields.items():
            parts.append(f"{k}={list(v.shape)}" if isinstance(v, torch.Tensor) else f"{k}=[{len(v)}]")
        return f"CrystalBatch(num_graphs={self.num_graphs}, {', '.join(parts)})"


_EDGE_FIELDS = ("edge_vec", "edge_attr", "edge_shift", "edge_len")
_GRAPH_CAT_FIELDS = ("glob", "y_ft", "y")            # 1-D per crystal, concatenated (`mat2graph.py:84-93`)
_GRAPH_STACK_FIELDS = ("system",)                     # 0-D per crystal -> [B]


def collate(crystals: Iterable[Dict[str, object]], sort_edges: bool = True,
            n_max: Optional[int] = None) -> CrystalBatch:
    """Concatenate per-crystal dicts into one batch (counterpart of PyG collate).

    Per crystal: ``x [n,Fa]``, ``edge_index [2,e]`` (local ids), and any of
    ``edge_vec [e,3]`` / ``edge_attr [e,Fb]``, ``glob [2]``, ``system`` scalar,
    ``phdos [1,51]``, ``y_ft [201]``, ``mp_id`` str (`utils.py:291-301`,
    `mat2graph.py:81-107`).  ``n_max`` may be forced larger than this batch's own
    maximum: data-parallel shards must pad to the GLOBAL batch's maximum to
    reproduce the single-process result.
    """
    crystals = list(crystals)
    if not crystals:
        raise ValueError("collate() needs at least one crystal")
    xs, eis, bvec, offs = [], [], [], 0
    for b, c in enumerate(crystals):
        n = c["x"].shape[0]
        xs.append(c["x"])
        eis.append(c["edge_index"].to(torch.int64) + offs)
        bvec.append(torch.full((n,), b, dtype=torch.int64))
        offs += n
    fields: Dict[str, object] = {
        "x": torch.cat(xs, 0),
        "edge_index": torch.cat(eis, 1),
        "batch": torch.cat(bvec, 0),
    }
    for k in _EDGE_FIELDS:
        if k in crystals[0] and isinstance(crystals[0][k], torch.Tensor):
            fields[k] = torch.cat([c[k] for c in crystals], 0)
    for k in _GRAPH_CAT_FIELDS:
        if k in crystals[0]:
            fields[k] = torch.cat([c[k].reshape(-1) for c in crystals], 0)
    for k in _GRAPH_STACK_FIELDS:
        if k in crystals[0]:
            fields[k] = torch.stack([torch.as_tensor(c[k]).reshape(()) for c in crystals]).to(torch.int64)
    if "phdos" in crystals[0]:
        fields["phdos"] = torch.cat([c["phdos"].reshape(1, -1) for c in crystals], 0)
    if "mp_id" in crystals[0]:
        fields["mp_id"] = [c["mp_id"] for c in crystals]

    ei = fields["edge_index"].numpy()
    meta = _build_meta_host(ei, fields["batch"].numpy(), len(crystals), n_max, presorted=False)
    if sort_edges and meta.edge_perm is not None:
        p = meta.edge_perm
        fields["edge_index"] = fields["edge_index"][:, p]
        for k in _EDGE_FIELDS:
            if k in fields:
                fields[k] = fields[k][p]
        meta.edge_perm = None
    return CrystalBatch(fields, len(crystals), meta)


def graph_meta(g, device=None, n_max: Optional[int] = None) -> GraphMeta:
    """Return (and cache on ``g``) the GraphMeta of a batch object.

    For a :class:`CrystalBatch` built by :func:`collate` this is free.  For a
    foreign object (e.g. a PyG ``Batch``) the metadata is derived from
    ``g.edge_index`` / ``g.batch`` with one device->host copy of the index
    arrays — the same host round trip the reference pays in ``to_dense_batch``.
    """
    m = getattr(g, _META_KEY, None) if not isinstance(g, CrystalBatch) else g.meta
    if (m is None or (n_max is not None and m.n_max != n_max)) and torch.is_tensor(g.edge_index) and g.edge_index.is_cuda:
        # foreign batch already on the GPU (the reference's loop: `batch.to(device)` then `model(batch)`,
        # main_eDOS.py:106-109): derive everything on the device; the only host read is the 4-byte n_max, and not
        # even that when the caller supplies it (the reference syncs on it inside to_dense_batch)
        m = graph_meta_device(g, n_max)
        try:
            if isinstance(g, CrystalBatch):
                object.__setattr__(g, _META_KEY, m)
            else:
                setattr(g, _META_KEY, m)
        except Exception:
            pass
        return m
    if m is None or (n_max is not None and m.n_max != n_max):
        ei = g.edge_index.detach().cpu().numpy()
        bv = g.batch.detach().cpu().numpy()
        nb = int(g.system.shape[0]) if hasattr(g, "system") and torch.is_tensor(g.system) and g.system.dim() > 0 \
            else (int(bv.max()) + 1 if bv.size else 0)
        presorted = bool(ei.shape[1] == 0 or np.all(ei[1, 1:] >= ei[1, :-1]))
        m = _build_meta_host(ei, bv, nb, n_max, presorted)
        try:
            if isinstance(g, CrystalBatch):
                object.__setattr__(g, _META_KEY, m)
            else:
                setattr(g, _META_KEY, m)
        except Exception:
            pass
    if device is not None and m.src.device != torch.device(device):
        m = m.to(device)
        try:
            if isinstance(g, CrystalBatch):
                object.__setattr__(g, _META_KEY, m)
            else:
                setattr(g, _META_KEY, m)
        except Exception:
            pass
    return m


def graph_meta_device(g, n_max: Optional[int] = None, num_graphs: Optional[int] = None) -> GraphMeta:
    """GraphMeta of a device-resident batch built by ``dosx_csr_build``: stable destination sort,
    CSR pointers, dense slots — no host round trip apart from the optional read of ``n_max``."""
    from . import ops
    ei = g.edge_index.to(torch.int64)
    bv = g.batch.to(torch.int64)
    if num_graphs is None:
        if isinstance(g, CrystalBatch):
            num_graphs = g.num_graphs
        elif hasattr(g, "system") and torch.is_tensor(g.system) and g.system.dim() > 0:
            num_graphs = int(g.system.shape[0])
        elif hasattr(g, "num_graphs"):
            num_graphs = int(g.num_graphs)
        else:
            num_graphs = int(bv[-1]) + 1 if bv.numel() else 0          # (host sync; PyG batches carry num_graphs)
    r = ops.csr_build(ei, bv, num_graphs)
    true_max = None
    if n_max is None:
        n_max = true_max = int(r["n_max"].item())
    return GraphMeta(num_nodes=int(bv.shape[0]), num_edges=int(ei.shape[1]), num_graphs=int(num_graphs), n_max=int(n_max),
                     src=r["src"], dst=r["dst"], edge_perm=r["edge_perm"], rowptr_dst=r["rowptr_dst"],
                     perm_src=r["perm_src"], rowptr_src=r["rowptr_src"], graph_ptr=r["graph_ptr"],
                     node_graph=r["node_graph"], dense_row=r["dense_row"], inv_deg=r["inv_deg"])


def split_crystals(g: CrystalBatch) -> List[Dict[str, object]]:
    """Inverse of :func:`collate` (used by the data-parallel sharder)."""
    bv = g.batch.cpu()
    ei = g.edge_index.cpu()
    counts = torch.bincount(bv, minlength=g.num_graphs)
    ptr = torch.zeros(g.num_graphs + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(counts, 0)
    egraph = bv[ei[1]] if ei.shape[1] else ei[1]
    out = []
    for b in range(g.num_graphs):
        sel = (egraph == b).nonzero().reshape(-1)
        c: Dict[str, object] = {
            "x": g.x[ptr[b]:ptr[b + 1]].cpu(),
            "edge_index": ei[:, sel] - ptr[b],
        }
        for k in _EDGE_FIELDS:
            if k in g and isinstance(g[k], torch.Tensor):
                c[k] = g[k].cpu()[sel]
        if "glob" in g:
            c["glob"] = g.glob.cpu().reshape(g.num_graphs, -1)[b]
        if "y_ft" in g:
            c["y_ft"] = g.y_ft.cpu().reshape(g.num_graphs, -1)[b]
        if "system" in g:
            c["system"] = g.system.cpu()[b]
        if "phdos" in g:
            c["phdos"] = g.phdos.cpu()[b:b + 1]
        if "mp_id" in g:
            c["mp_id"] = g.mp_id[b]
        out.append(c)
    return out


# --------------------------------------------------------------------------------------------------
# Shape bucketing for HIP-graph replay: pad a (destination-sorted) batch with GHOST nodes / edges.
# --------------------------------------------------------------------------------------------------
def bucket_sizes(num_nodes: int, num_edges: int, node_step: int = 8, edge_step: int = 128):
    """Padded (N, E) of the bucket a batch falls into; always leaves room for >= 1 ghost node."""
    n_pad = (num_nodes + 1 + node_step - 1) // node_step * node_step
    e_pad = (num_edges + edge_step - 1) // edge_step * edge_step
    return n_pad, e_pad


def pad_seg_tiles(t: Optional[torch.Tensor], N: int, E: int, n_pad: int, e_pad: int, B: int) -> Optional[torch.Tensor]:
    """Tile table of a ghost-padded batch, [3, T+1] with T = seg_tile_bound(...) slots: the real tiles, then the ghost edges
    in SEG_TILE_ROWS-row tiles (the first of them also owns every ghost node - their aggregate is a finite don't-care),
    then empty slots.  Mirrors what dosx_collate_padded writes on the device."""
    if t is None:
        return None
    T = seg_tile_bound(n_pad, e_pad, B)
    real = int(t.shape[1]) - 1
    ng = max(1, (e_pad - E + SEG_TILE_ROWS - 1) // SEG_TILE_ROWS)
    if real + ng > T:
        return None
    k = torch.arange(T + 1 - real, dtype=torch.int64, device=t.device)
    eb = torch.clamp(E + k * SEG_TILE_ROWS, max=e_pad)
    nb = torch.where(k == 0, torch.full_like(k, N), torch.full_like(k, n_pad))
    tail = torch.stack([eb, nb, torch.zeros_like(k)]).to(torch.int32)
    return torch.cat([t[:, :real], tail], 1).contiguous()


def pad_batch(g: CrystalBatch, n_pad: int, e_pad: int) -> CrystalBatch:
    """Pad to fixed (N, E) so that every kernel launch of a training step has static geometry and the
    step can be replayed from a captured HIP graph.

    Ghost nodes carry zero features, belong to NO crystal (outside every ``graph_ptr`` range, their
    ``node_graph`` / ``dense_row`` point at one spare zero row past the real ones) and ghost edges
    are self loops on the first ghost node.  Nothing a real crystal computes reads a ghost row, and every
    gradient reaching a ghost row is exactly zero, so outputs and parameter gradients are bitwise
    those of the unpadded batch (tests/test_gpu_models.py::test_ghost_padding_is_exact)."""
    m = g.meta
    if m is None or m.edge_perm is not None:
        raise ValueError("pad_batch needs a batch built by collate(sort_edges=True)")
    N, E, B = m.num_nodes, m.num_edges, m.num_graphs
    if n_pad < N + 1 or e_pad < E:
        raise ValueError(f"bucket ({n_pad},{e_pad}) too small for N={N}, E={E} (+1 ghost node)")
    dn, de = n_pad - N, e_pad - E
    f = dict(g._fields)
    dev = g.x.device

    def padrows(t, n):
        return torch.cat([t, t.new_zeros((n,) + tuple(t.shape[1:]))], 0) if n > 0 else t

    f["x"] = padrows(g.x, dn)
    f["batch"] = torch.cat([g.batch, torch.full((dn,), B, dtype=g.batch.dtype, device=dev)])
    f["edge_index"] = torch.cat([g.edge_index, torch.full((2, de), N, dtype=g.edge_index.dtype, device=dev)], 1)
    for k in _EDGE_FIELDS:
        if k in f and isinstance(f[k], torch.Tensor):
            f[k] = padrows(f[k], de)
    i32 = lambda v, n: torch.full((n,), v, dtype=torch.int32, device=m.src.device)
    deg_ghost = torch.ones(dn, dtype=torch.float32, device=m.inv_deg.device)
    deg_ghost[0] = 1.0 / max(de, 1)
    meta = GraphMeta(
        num_nodes=n_pad, num_edges=e_pad, num_graphs=B, n_max=m.n_max,
        src=torch.cat([m.src, i32(N, de)]), dst=torch.cat([m.dst, i32(N, de)]), edge_perm=None,
        rowptr_dst=torch.cat([m.rowptr_dst, i32(e_pad, dn)]),
        perm_src=torch.cat([m.perm_src, torch.arange(E, e_pad, dtype=torch.int32, device=m.perm_src.device)]),
        rowptr_src=torch.cat([m.rowptr_src, i32(e_pad, dn)]),
        graph_ptr=m.graph_ptr, node_graph=torch.cat([m.node_graph, i32(B, dn)]),
        dense_row=torch.cat([m.dense_row, i32(m.n_max * B, dn)]),
        inv_deg=torch.cat([m.inv_deg, deg_ghost]),
        seg_tile=pad_seg_tiles(m.seg_tile, N, E, n_pad, e_pad, B),
    )
    out = CrystalBatch(f, B, meta)
    object.__setattr__(out, "real_nodes", N)
    object.__setattr__(out, "n_global", g.n_global)
    return out
