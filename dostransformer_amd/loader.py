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
from .batch import CrystalBatch, GraphMeta, _EDGE_FIELDS


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
            rps.append(np.concatenate([[0], np.cumsum(deg_out)]))
            invd.append((1.0 / np.maximum(deg_in, 1)).astype(np.float32))
            for k in _EDGE_FIELDS:
                if k in c and isinstance(c[k], torch.Tensor):
                    edge_feats[k].append(c[k][torch.from_numpy(order)])
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(np.concatenate(a).astype(np.int32))).to(self.device)
        self._src, self._dst, self._perm = i32(srcs), i32(dsts), i32(perms)
        self._rpd, self._rps = i32(rpd), i32(rps)
        self._invd = torch.from_numpy(np.concatenate(invd)).to(self.device)
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
        meta = GraphMeta(num_nodes=N, num_edges=E, num_graphs=B, n_max=int(n_max), edge_perm=None, graph_ptr=onp, **m)
        return CrystalBatch(fields, B, meta)

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
