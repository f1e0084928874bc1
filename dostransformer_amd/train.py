"""Fused training step of the reference callers (`main_phDOS.py:101-118`, `main_eDOS.py:101-127`):
forward program -> loss kernel -> backward program -> (RCCL all-reduce) -> flat AdamW kernel.

No autograd graph, no per-parameter optimizer loop, no host synchronisation: the loss stays on the
device (the reference prints it every step, `main_eDOS.py:129`; read ``.item()`` only when needed).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as Fn
from . import ops
from ._models import DOSTransformerBase
from .batch import graph_meta


class Trainer:
    """AdamW(lr, weight_decay=1e-2) training of a DOSTransformer(_phonon) module, all on libdosx.

    ``dist``: optional :class:`dostransformer_amd.dist.DataParallel` — shards are per-rank batches,
    gradients are summed over ranks (loss kernels already divide by the GLOBAL element/crystal
    count), the phonon loss exchanges its two SSE scalars before the backward pass.
    """

    def __init__(self, model: DOSTransformerBase, lr: float = 1e-4, beta: float = 1.0, weight_decay: float = 1e-2,
                 betas=(0.9, 0.999), eps: float = 1e-8, dist=None):
        if not isinstance(model, DOSTransformerBase):
            raise TypeError("Trainer drives DOSTransformer / DOSTransformer_phonon modules")
        self.model, self.lr, self.beta, self.wd, self.betas, self.eps = model, lr, beta, weight_decay, betas, eps
        self.dist = dist
        self.step_count = 0
        self._m = self._v = None
        self._fp = None
        self.kind = model._cfg.kind
        self.last_outputs = None

    def _state(self, fp):
        if self._fp is not fp:
            self._m = torch.zeros_like(fp.flat)
            self._v = torch.zeros_like(fp.flat)
            self._fp = fp
        return self._m, self._v

    def forward_backward(self, g, n_global: Optional[int] = None) -> torch.Tensor:
        """Forward + loss + backward; leaves the gradients in the flat buffer.  Returns the loss
        (0-dim device tensor; for eDOS under data parallelism it is this rank's share)."""
        model = self.model
        dev = model._module_device()
        fp = model._ensure_flat(dev, g)
        m = graph_meta(g, dev)
        cfg = model._cfg
        B, S = m.num_graphs, cfg.S
        with torch.no_grad():
            dg, xL, ds, (ctx, dos) = model._program_fwd(fp.P, g, m)
            self.last_outputs = (dg, xL, ds)
            ddos = torch.empty_like(dos)
            if self.kind == "phonon":
                y = Fn._f32(g.phdos).reshape(B, S)
                sse = torch.empty(2, device=dev, dtype=torch.float32)
                ops.sse2(dos[:B], dos[B:], y, sse, B * S)
                count_global = float(B * S)
                if self.dist is not None:
                    count_global = float(self.dist.all_reduce_sum_scalars(sse, B * S))
                loss = torch.empty(1, device=dev, dtype=torch.float32)
                ops.loss_phonon_bwd(dos[:B], dos[B:], y, sse, self.beta, count_global, ddos[:B], ddos[B:], loss,
                                    B * S)
                loss = loss[0]
            else:
                y = Fn._f32(g.y_ft).reshape(-1)
                bg = B if n_global is None else n_global
                if self.dist is not None and n_global is None:
                    bg = self.dist.global_count(B)
                lp = torch.empty(B, device=dev, dtype=torch.float32)
                ops.loss_edos(dos[:B], dos[B:], y, self.beta, B, S, bg, ddos[:B], ddos[B:], lp)
                loss = lp.sum()
            sink = ops.GradSink(dev)
            Fn.dostransformer_bwd(fp.P, fp.G, cfg, m, ctx, ddos, None, sink)
            sink.release()
        return loss

    def optimizer_step(self) -> None:
        fp = self._fp if self._fp is not None else self.model.flat_params()
        m, v = self._state(fp)
        if self.dist is not None:
            self.dist.all_reduce_grads(fp.grad)
        self.step_count += 1
        ops.adamw(fp.flat, fp.grad, m, v, fp.total, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                  self.step_count, 1.0)

    def step(self, g, n_global: Optional[int] = None) -> torch.Tensor:
        fp = self.model.flat_params(g)
        self._state(fp)
        loss = self.forward_backward(g, n_global)
        self.optimizer_step()
        return loss
