"""Device-resident dataset + collate on the GPU (counterpart of the PyG ``DataLoader`` the drivers
build at `main_eDOS.py:54` / `main_phDOS.py:53` and of the per-step ``batch.to(device)`` at `main_eDOS.py:106`).

The reference collates on the host every step and copies the batch to the GPU.  Here every crystal is uploaded ONCE,
with its edges already sorted by destination and its local CSR cached; a batch is then assembled on the device by
segment copies with the batch's node / edge offsets added (``dosx_collate``) plus row gathers of the feature tensors.
Per step the host only computes two prefix sums over the B selected crystals and ships 3 small index arrays.
The result is bit-identical to ``batch.collate([crystals[i] for i in indices])``
(tests/test_gpu_models.py::test_device_collate_matches_host)."""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .batch import SEG_TILE_ROWS, CrystalBatch, GraphMeta, _EDGE_FIELDS, seg_tile_bound, seg_tiles_host


class DeviceDataset:
    def __init__(self, crystals: Sequence[Dict[str, object]], device, dtype: Optional[torch.dtype] = None):
        crystals = list(crystals)
        if not crystals:
            raise ValueError("DeviceDataset needs at least one crystal")
        self.device = torch.device(device)
        C = len(crystals)
        n_nodes = np.array([int(c["x"].shape[0]) for c in crystals], np.int64)
        n_edges = np.array([int(c["edge_index"].shape[1]) for c in crystals], np.int64)
        self.n_nodes, self.n_edges = n_nodes, n_edges
        self.node_ptr = np.concatenate([[0], np.cumsum(n_nodes)]).astype(np.int64)
        self.edge_ptr = np.concatenate([[0], np.cumsum(n_edges)]).astype(np.int64)
        srcs, dsts, perms, rpd, rps, invd, edge_feats = [], [], [], [], [], [], {k: [] for k in _EDGE_FIELDS}
        tiles_e, tiles_n, tiles_p, tile_cnt = [], [], [], []     # per-crystal node-aligned row tiles (message GEMM, EPI_SEGSUM)
        for c in crystals:
            ei = c["edge_index"].numpy().astype(np.int64)
            n = int(c["x"].shape[0])
            order = np.argsort(ei[1], kind="stable")                      # destination-sorted, like batch.collate
            s, d = ei[0][order], ei[1][order]
            deg_in = np.bincount(d, minlength=n)
            deg_out = np.bincount(s, minlength=n)
            srcs.append(s); dsts.append(d)
            perms.append(np.argsort(s, kind="stable"))
            rpd.append(np.concatenate([[0], np.cumsum(deg_in)]))
            tl = seg_tiles_host(rpd[-1])
            tiles_e.append(tl[0][:-1]); tiles_n.append(tl[1][:-1]); tiles_p.append(tl[2][:-1]); tile_cnt.append(tl.shape[1] - 1)
            rps.append(np.concatenate([[0], np.cumsum(deg_out)]))
            invd.append((1.0 / np.maximum(deg_in, 1)).astype(np.float32))
            for k in _EDGE_FIELDS:
                if k in c and isinstance(c[k], torch.Tensor):
                    edge_feats[k].append(c[k][torch.from_numpy(order)])
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate(a).astype(np.int32))).to(self.device)
        self._src, self._dst, self._perm = i32(srcs), i32(dsts), i32(perms)
        self._rpd, self._rps = i32(rpd), i32(rps)
        self._invd = torch.from_numpy(np.concatenate(invd)).to(self.device)
        self.tile_cnt = np.asarray(tile_cnt, np.int64)
        self._tiles_host = (tiles_e, tiles_n, tiles_p)            # (collate() assembles a batch's table on the host)
        self._tile_e, self._tile_n, self._tile_p = i32(tiles_e), i32(tiles_n), i32(tiles_p)
        self._tile_off = torch.from_numpy(np.concatenate([[0], np.cumsum(self.tile_cnt)]).astype(np.int32)).to(self.device)
        self._node_ptr = torch.from_numpy(self.node_ptr.astype(np.int32)).to(self.device)
        self._edge_ptr = torch.from_numpy(self.edge_ptr.astype(np.int32)).to(self.device)
        f = lambda t: (t.to(dtype) if dtype is not None and t.is_floating_point() else t).to(self.device)
        self._x = f(torch.cat([c["x"] for c in crystals], 0))
        self._edge = {k: f(torch.cat(v, 0)) for k, v in edge_feats.items() if v}
        self._graph: Dict[str, torch.Tensor] = {}
        for k in ("glob", "y_ft", "y", "phdos"):
            if k in crystals[0]:
                self._graph[k] = f(torch.stack([c[k].reshape(-1) for c in crystals], 0))
        if "system" in crystals[0]:
            self._graph["system"] = torch.stack([torch.as_tensor(c["system"]).reshape(()) for c in crystals]).to(torch.int64).to(self.device)
        self.mp_id = [c["mp_id"] for c in crystals] if "mp_id" in crystals[0] else None
        self.num_crystals = C

    def __len__(self) -> int:
        return self.num_crystals

    def collate(self, indices: Sequence[int], n_max: Optional[int] = None) -> CrystalBatch:
        idx = np.asarray(list(indices), np.int64)
        B = int(idx.shape[0])
        nn, ne = self.n_nodes[idx], self.n_edges[idx]
        out_np = np.concatenate([[0], np.cumsum(nn)]).astype(np.int32)
        out_ep = np.concatenate([[0], np.cumsum(ne)]).astype(np.int32)
        N, E = int(out_np[-1]), int(out_ep[-1])
        true_max = int(nn.max())
        if n_max is None:
            n_max = true_max
        elif n_max < true_max:
            raise ValueError(f"n_max={n_max} smaller than the largest crystal ({true_max} atoms)")
        dev = self.device
        small = torch.from_numpy(np.concatenate([idx.astype(np.int32), out_np, out_ep])).to(dev, non_blocking=True)
        sel, onp, oep = small[:B], small[B:2 * B + 1], small[2 * B + 1:]
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
        batch = torch.empty(N, dtype=torch.int64, device=dev)
        edge_index = torch.empty(2, E, dtype=torch.int64, device=dev)
        m = dict(src=i32(E), dst=i32(E), perm_src=i32(E), rowptr_dst=i32(N + 1), rowptr_src=i32(N + 1), node_graph=i32(N),
                 dense_row=i32(N), inv_deg=torch.empty(N, dtype=torch.float32, device=dev))
        node_row, edge_row = i32(N), i32(E)
        ops._call("dosx_collate", sel.data_ptr(), self._node_ptr.data_ptr(), self._edge_ptr.data_ptr(), onp.data_ptr(),
                  oep.data_ptr(), B, N, E, self._src.data_ptr(), self._dst.data_ptr(), self._perm.data_ptr(),
                  self._rpd.data_ptr(), self._rps.data_ptr(), self._invd.data_ptr(), batch.data_ptr(), edge_index.data_ptr(),
                  m["src"].data_ptr(), m["dst"].data_ptr(), m["perm_src"].data_ptr(), m["rowptr_dst"].data_ptr(),
                  m["rowptr_src"].data_ptr(), m["node_graph"].data_ptr(), m["dense_row"].data_ptr(), m["inv_deg"].data_ptr(),
                  node_row.data_ptr(), edge_row.data_ptr(), ops._stream())
        fields: Dict[str, object] = {"x": self._x.index_select(0, node_row), "edge_index": edge_index, "batch": batch}
        for k, v in self._edge.items():
            fields[k] = v.index_select(0, edge_row)
        sel64 = sel.to(torch.int64)
        for k, v in self._graph.items():
            g = v.index_select(0, sel64)
            fields[k] = g if k in ("phdos", "system") else g.reshape(-1)
        if self.mp_id is not None:
            fields["mp_id"] = [self.mp_id[i] for i in idx]
        # the message GEMM's tile table: the selected crystals' tables with the batch offsets added - the table
        # dosx_collate_padded builds for step_dataset(), so that step(collate(...)) takes the same kernels and buckets
        te, tn, tp = self._tiles_host
        seg = np.concatenate([np.stack([te[c] + out_ep[b], tn[c] + out_np[b], tp[c]]) for b, c in enumerate(idx)] +
                             [np.array([[E], [N], [0]], np.int32)], 1).astype(np.int32)
        meta = GraphMeta(num_nodes=N, num_edges=E, num_graphs=B, n_max=int(n_max), edge_perm=None, graph_ptr=onp,
                         seg_tile=torch.from_numpy(np.ascontiguousarray(seg)).to(dev, non_blocking=True), **m)
        return CrystalBatch(fields, B, meta)

    # ---- collate straight into a shape bucket's static buffers (train.Trainer.step_dataset) -------------------------
    def _f32_tables(self):
        """fp32 / int32 device copies of the feature tables in the kernels' format (built once, on first use)."""
        t = getattr(self, "_tables32", None)
        if t is None:
            ek = "edge_vec" if "edge_vec" in self._edge else "edge_attr"
            tk = "phdos" if "phdos" in self._graph else "y_ft"
            t = {"x": self._x.to(torch.float32).contiguous(), "edge": self._edge[ek].to(torch.float32).contiguous(),
                 "edge_key": ek, "target": self._graph[tk].to(torch.float32).contiguous(), "target_key": tk,
                 "glob": self._graph["glob"].to(torch.float32).contiguous() if "glob" in self._graph else None,
                 "system": self._graph["system"].to(torch.int32).contiguous()}
            self._tables32 = t
        return t

    def bucket_dims(self, indices, n_max: Optional[int] = None):
        """(idx, N, E, n_max) of a selection, from host-side counts only."""
        idx = np.asarray(list(indices), np.int64)
        nn = self.n_nodes[idx]
        true_max = int(nn.max())
        if n_max is None:
            n_max = true_max
        elif n_max < true_max:
            raise ValueError(f"n_max={n_max} smaller than the largest crystal ({true_max} atoms)")
        return idx, int(nn.sum()), int(self.n_edges[idx].sum()), int(n_max)

    def collate_into(self, g: CrystalBatch, idx: np.ndarray, scratch: Dict[str, torch.Tensor]) -> None:
        """Write the batch of crystals ``idx`` into the STATIC ghost-padded buffers of ``g`` (a bucket of
        ``train.Trainer``): one small host->device copy (selection + prefix sums) and one ``dosx_collate_padded`` call
        (3 launches).  The result is what ``pad_batch(self.collate(idx), N_pad, E_pad)`` holds in the fields the kernels read
        (tests/test_gpu_graph.py::test_collate_into_matches_pad_batch)."""
        from ._lib import Collate
        import ctypes as C
        t = self._f32_tables()
        m = g.meta
        B = int(idx.shape[0])
        nn, ne = self.n_nodes[idx], self.n_edges[idx]
        out_np = np.concatenate([[0], np.cumsum(nn)]).astype(np.int32)
        out_ep = np.concatenate([[0], np.cumsum(ne)]).astype(np.int32)
        small = scratch["small"]
        parts = [idx.astype(np.int32), out_np, out_ep]
        tiled = m.seg_tile is not None
        if tiled:
            parts.append(np.concatenate([[0], np.cumsum(self.tile_cnt[idx])]).astype(np.int32))
        host = torch.from_numpy(np.concatenate(parts))
        small[:host.numel()].copy_(host, non_blocking=True)
        d = Collate()
        d.B, d.N, d.E, d.N_pad, d.E_pad, d.n_max = B, int(out_np[-1]), int(out_ep[-1]), m.num_nodes, m.num_edges, m.n_max
        d.Fa, d.Fe, d.S = int(t["x"].shape[1]), int(t["edge"].shape[1]), int(t["target"].shape[1])
        d.n_glob = int(t["glob"].shape[1]) if t["glob"] is not None else 0
        base = small.data_ptr()
        d.sel, d.out_node_ptr, d.out_edge_ptr = base, base + 4 * B, base + 4 * (2 * B + 1)
        d.node_ptr_all, d.edge_ptr_all = self._node_ptr.data_ptr(), self._edge_ptr.data_ptr()
        d.src_all, d.dst_all, d.perm_src_all = self._src.data_ptr(), self._dst.data_ptr(), self._perm.data_ptr()
        d.rowptr_dst_all, d.rowptr_src_all, d.inv_deg_all = self._rpd.data_ptr(), self._rps.data_ptr(), self._invd.data_ptr()
        d.x_all, d.edge_feat_all, d.target_all = t["x"].data_ptr(), t["edge"].data_ptr(), t["target"].data_ptr()
        d.glob_all = t["glob"].data_ptr() if t["glob"] is not None else None
        d.system_all = t["system"].data_ptr()
        d.x, d.edge_feat, d.target = g.x.data_ptr(), g[t["edge_key"]].data_ptr(), g[t["target_key"]].data_ptr()
        d.glob = g.glob.data_ptr() if t["glob"] is not None else None
        d.system = g.system.data_ptr()
        for k in ("src", "dst", "perm_src", "rowptr_dst", "rowptr_src", "graph_ptr", "node_graph", "dense_row", "inv_deg"):
            setattr(d, k, getattr(m, k).data_ptr())
        d.node_row, d.edge_row = scratch["node_row"].data_ptr(), scratch["edge_row"].data_ptr()
        if tiled:
            d.T, d.tile_rows = int(m.seg_tile.shape[1]) - 1, SEG_TILE_ROWS
            if int(parts[3][-1]) + max(1, -(-(d.E_pad - d.E) // SEG_TILE_ROWS)) > d.T:
                raise ValueError("tile table of the bucket is too small for this batch")
            d.out_tile_ptr = base + 4 * (3 * B + 2)
            d.tile_off_all, d.tile_e_all, d.tile_n_all = self._tile_off.data_ptr(), self._tile_e.data_ptr(), self._tile_n.data_ptr()
            d.tile_p_all = self._tile_p.data_ptr()
            d.seg_tile = m.seg_tile.data_ptr()
        ops._call("dosx_collate_padded", C.byref(d), ops._stream(),
                  w=lambda: ("collate_padded", "collate_pad", "hbm", 8.0 * (d.N_pad * d.Fa + d.E_pad * d.Fe)))

    def batches(self, batch_size: int, shuffle: bool = False, seed: int = 0, drop_last: bool = False) -> Iterator[CrystalBatch]:
        """One epoch of device-collated batches (the loop body of `main_eDOS.py:104`)."""
        order = np.arange(self.num_crystals)
        if shuffle:
            np.random.default_rng(seed).shuffle(order)
        for i in range(0, self.num_crystals, batch_size):
            sel = order[i:i + batch_size]
            if drop_last and len(sel) < batch_size:
                break
            yield self.collate(sel)
