"""Replayed inference (small-batch / batch-1 prediction is launch-bound).

``Predictor(model)(batch)`` returns what ``model(batch)`` returns under ``torch.no_grad()`` —
(dos_global, x, dos_system), `DOSTransformer_phonon.py:66-119` / `DOSTransformer.py` forward — but issues the
forward program from a recorded launch list (``ops.Program`` -> ``dosx_replay``) on static buffers of the
batch's (N, E, B, n_max) bucket: the first call on a bucket runs eagerly while recording, later calls are
"copy the batch in, replay".  Ghost padding is exact (``batch.pad_batch``), so the outputs are bitwise those
of the eager forward.  The returned tensors alias the bucket's buffers and are overwritten by the next call
that lands in the same bucket; ``.clone()`` them to keep them.

It quacks like the module for the evaluation loops: ``evaluate.test(Predictor(model), loader)``.

``Predictor(model, per_crystal_keys=True)``: the reference validates and tests at ``batch_size = 1`` (`main_eDOS.py:55-56`,
`utils.py:61-143`), and its outputs depend on the batch's Nmax (the zero-padded atoms take part in the softmax,
`DOSTransformer_phonon.py:86`), so reference metrics need batch-1 forwards - 300 us each, launch-bound.  With this flag the two
cross attentions of a BATCHED forward attend over each crystal's own atoms only (``DosxAttn.key_ptr`` = the batch's graph_ptr):
B crystals in one pass give what B batch-1 forwards give (to fp32 rounding), at the batched rate.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from ._models import DOSTransformerBase
from .batch import CrystalBatch, bucket_sizes, graph_meta, pad_batch
from .train import _Slot


class Predictor:
    def __init__(self, model: DOSTransformerBase, bucket=(8, 128), per_crystal_keys: bool = False):
        if not isinstance(model, DOSTransformerBase):
            raise TypeError("Predictor drives DOSTransformer / DOSTransformer_phonon modules")
        self.model = model
        self.bucket = tuple(bucket)
        self.per_crystal_keys = bool(per_crystal_keys)
        self.kind = model._cfg.kind
        self._fp = None
        self._slots: Dict[tuple, _Slot] = {}

    def eval(self):
        self.model.eval()
        return self

    def _record(self, slot: _Slot, fp) -> None:
        timer_on = ops.KERNEL_TIMER.enabled
        ops.KERNEL_TIMER.enabled = False
        g = slot.g
        try:
            with torch.no_grad():
                ops.RECORDER.begin()
                dg, xL, ds, keep = self.model._program_fwd(fp.P, g, g.meta, per_crystal_keys=self.per_crystal_keys)
                slot.prog_a = ops.RECORDER.end()
        finally:
            if ops.RECORDER.active:
                ops.RECORDER.end()
            ops.KERNEL_TIMER.enabled = timer_on
        slot.keep = keep                      # the program's intermediates live as long as the recording
        slot.out = (dg, xL, ds)

    def __call__(self, g: CrystalBatch):
        model = self.model
        if model.training and getattr(model, "_attn_drop", 0.0) > 0.0:
            raise RuntimeError("Predictor replays an inference program: call model.eval() first (attention dropout is "
                               "active in training mode)")
        dev = model._module_device()
        if dev.type != "cuda":
            raise RuntimeError("Predictor runs only on an MI355X through libdosx (no CPU fallback)")
        fp = model._ensure_flat(dev, g)
        if fp is not self._fp:                # parameters were re-homed: recorded pointers are stale
            self._fp, self._slots = fp, {}
        m = graph_meta(g, dev)
        if m.edge_perm is not None:
            raise ValueError("Predictor needs batches from collate(sort_edges=True) / DeviceDataset.collate")
        n_real = getattr(g, "real_nodes", None)
        if n_real is None:
            n_real = m.num_nodes
            cached = getattr(g, "_dosx_padded", None)          # evaluation loops revisit the same batch objects:
            if cached is None or cached[0] != self.bucket:     # pad (≈20 small torch ops) only once per batch
                cached = (self.bucket, pad_batch(g, *bucket_sizes(m.num_nodes, m.num_edges, *self.bucket)))
                try:
                    object.__setattr__(g, "_dosx_padded", cached)
                except (AttributeError, TypeError):
                    pass
            g = cached[1]
            m = g.meta
        key = (m.num_nodes, m.num_edges, m.num_graphs, m.n_max)
        slot = self._slots.get(key)
        if slot is None:
            slot = _Slot(g, self.kind, targets=False)
            self._slots[key] = slot
            self._record(slot, fp)
        else:
            slot.load(g)
            slot.prog_a.run()
        dg, xL, ds = slot.out
        return dg, xL[:n_real], ds
