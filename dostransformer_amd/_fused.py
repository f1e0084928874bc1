"""Glue between ``torch.nn.Module`` parameter storage / autograd and the libdosx programs in
``functional.py``.

* live parameters are re-homed (lazily, on the device the module lives on) into ONE flat fp32
  buffer, with a same-shaped flat gradient buffer — the layout the fused AdamW kernel and the
  data-parallel RCCL all-reduce want.  ``state_dict()`` keys/shapes are unchanged.
* parameters the reference never uses (``self_attn.in_proj_*``, ``self_attn.out_proj.*``,
  ``node_mlp_1.*``, ``alpha`` — SURVEY.md §0.2/§0.6) stay outside: they get ``grad = None`` and are
  therefore skipped by AdamW exactly like upstream.
* one ``torch.autograd.Function`` spans the whole model: forward runs the forward program and keeps
  its activations, backward runs the backward program and publishes ``param.grad`` as views of the
  flat gradient buffer (no per-parameter copies).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .batch import GraphMeta, graph_meta

_DEAD = re.compile(r"(\.self_attn\.)|(\.node_mlp_1\.)|(^alpha$)")
_ALIGN = 64   # floats; keeps every parameter 256-B aligned inside the flat buffer


def is_dead_param(name: str) -> bool:
    return _DEAD.search(name) is not None


_LATE = ("GN_encoder.", "stacked_processor.", "GN_decoder.")
_LAST = ("GN_encoder.", "stacked_processor.0.")


def is_late_param(name: str) -> bool:
    """Parameters of the GNN trunk: their gradients are complete only at the very end of the backward pass.  All
    others (transformer stacks, heads, embeddings) are final once the backward reaches the GNN, so their slice of
    the flat gradient buffer can be all-reduced while the GNN backward still runs (dist / train.Trainer)."""
    return name.startswith(_LATE)


def is_last_param(name: str) -> bool:
    """The part of the GNN trunk whose gradients exist only at the END of the backward pass: the node / edge / global encoders and
    message-passing layer 0.  The rest of the trunk (GN_decoder, layers 1 .. L-1: the MID bucket) is final once the backward reaches
    layer 0 - its all-reduce starts there, under layer 0's backward, and only this bucket stays exposed behind the step."""
    return name.startswith(_LAST)


class FlatParams:
    """Flat fp32 parameter + gradient storage for the live parameters of a module."""

    def __init__(self, module: nn.Module, device: torch.device, extra_dead=()):
        live = [(n, p) for n, p in module.named_parameters() if not (is_dead_param(n) or n in extra_dead)]
        # layout: [ last bucket (encoders + layer 0) | mid bucket (rest of the GNN trunk) | early bucket (everything else) ],
        # each in module order; [last | mid] together are the "late" slice flat[:n_late]
        ordered = [x for x in live if is_last_param(x[0])] + [x for x in live if is_late_param(x[0]) and not is_last_param(x[0])] + \
                  [x for x in live if not is_late_param(x[0])]
        names, params = [n for n, _ in ordered], [p for _, p in ordered]
        offs, tot, n_late, n_last = [], 0, 0, 0
        for n, p in ordered:
            offs.append(tot)
            tot += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if is_late_param(n):
                n_late = tot
            if is_last_param(n):
                n_last = tot
        self.names, self.offsets, self.total = names, offs, tot
        self.n_late = n_late          # floats: flat[:n_late] is the GNN trunk, flat[n_late:] the early bucket
        self.n_last = n_last          # floats: flat[:n_last] the last bucket, flat[n_last:n_late] the mid bucket
        self.flat = torch.zeros(tot, device=device, dtype=torch.float32)
        self.grad = torch.zeros(tot, device=device, dtype=torch.float32)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        self.params: Dict[str, nn.Parameter] = {}
        with torch.no_grad():
            for n, p, o in zip(names, params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                self.P[n] = view
                self.G[n] = self.grad[o:o + p.numel()].view(p.shape)
                self.params[n] = p
        for n, p in module.named_parameters():       # dead parameters just follow the module's device
            if n not in self.P and p.device != device:
                p.data = p.data.to(device)
        self.anchor = params[0]
        self.anchor_ptr = self.flat.data_ptr() + 4 * offs[0]

    def intact(self, device: torch.device) -> bool:
        return self.flat.device == device and self.anchor.data_ptr() == self.anchor_ptr and \
            self.anchor.dtype == torch.float32

    def publish_grads(self, G: Dict[str, torch.Tensor], accumulate_into_existing: bool) -> None:
        for n, p in self.params.items():
            if p.grad is None or not accumulate_into_existing:
                p.grad = G[n]
            else:
                p.grad = p.grad + G[n]


class _ModelFn(torch.autograd.Function):
    """Whole-model autograd node (see module docstring)."""

    @staticmethod
    def forward(ctx, anchor, model, g, m):
        fp: FlatParams = model._flat
        out = model._program_fwd(fp.P, g, m)
        ctx.model, ctx.m, ctx.saved, ctx.fp = model, m, out[-1], fp
        res = out[:-1]
        ctx.n_out = len(res)
        ctx.set_materialize_grads(False)
        return res

    @staticmethod
    def backward(ctx, *grads):
        model, fp = ctx.model, ctx.fp
        busy = any(p.grad is not None for p in fp.params.values())
        if busy:
            # true accumulation (two backward() calls without zero_grad): use a scratch gradient buffer
            gbuf = torch.zeros_like(fp.grad)
            G = {n: gbuf[o:o + fp.P[n].numel()].view(fp.P[n].shape) for n, o in zip(fp.names, fp.offsets)}
        else:
            G = fp.G
        sink = ops.GradSink(fp.flat.device)
        model._program_bwd(fp.P, G, ctx.m, ctx.saved, grads, sink)
        sink.release()
        fp.publish_grads(G, busy)
        ctx.saved = None
        return None, None, None, None


class FusedModel(nn.Module):
    """Base class of the drop-in model modules: lazily flattens parameters and routes forward /
    backward through the libdosx programs."""

    _flat: Optional[FlatParams] = None

    def _extra_dead(self, g) -> Tuple[str, ...]:
        return ()

    def _ensure_flat(self, device: torch.device, g) -> FlatParams:
        dead = self._extra_dead(g)
        fp = self._flat
        if fp is None or not fp.intact(device) or getattr(self, "_flat_dead", ()) != dead:
            fp = FlatParams(self, device, dead)
            object.__setattr__(self, "_flat", fp)
            object.__setattr__(self, "_flat_dead", dead)
        return fp

    def _module_device(self) -> torch.device:
        return next(self.parameters()).device

    def flat_params(self, g=None) -> FlatParams:
        return self._ensure_flat(self._module_device(), g)

    def _run(self, g):
        dev = self._module_device()
        if dev.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__} runs only on an MI355X through libdosx (no CPU fallback): "
                f"move the module with .to('cuda') first (it is on {dev}).")
        fp = self._ensure_flat(dev, g)
        m = graph_meta(g, dev)
        if torch.is_grad_enabled():
            out = _ModelFn.apply(fp.anchor, self, g, m)
        else:
            out = self._program_fwd(fp.P, g, m)[:-1]
        # The kernels compute in fp32.  A float64 batch (the phonon reference sets the default dtype to float64,
        # main_phDOS.py:15-16) gets float64 outputs back, like upstream: the caller's MSELoss against its float64 target
        # then back-propagates a float64 gradient, which this (autograd-aware) cast turns into the fp32 one the backward
        # program takes.
        dt = g.x.dtype if torch.is_tensor(getattr(g, "x", None)) else torch.float32
        if dt.is_floating_point and dt != torch.float32:
            out = tuple(t.to(dt) for t in out)
        return out
