"""Building blocks of the message-passing stack (`DOSTransformer_phonon.py:126-212`, `DOSTransformer.py:100-190`).

They reproduce the reference's module tree (hence its ``state_dict`` keys and, with the same seed, its initial values).
Inside the full models their parameters are read directly by the fused programs of ``functional.py``; called ON THEIR
OWN — ``Processor(x, edge_index, edge_attr)``, ``EdgeModel(src, dest, e)``, ``NodeModel(x, edge_index, e)``,
``Encoder(...)``, ``Decoder(...)`` with the upstream signatures — they run the same libdosx kernels through a small
autograd node (forward program / backward program), so the reference's block-level behaviour (isolated nodes, duplicate
edges, ``node_mlp_1`` never touched) is reproduced on the GPU as well (fixture G3).  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from . import ops
from .batch import GraphMeta, _build_meta_host
from .ops import seg


def mlp_prelu(n_in: int, n_hidden: int) -> nn.Sequential:
    """Linear -> PReLU -> Linear (the three encoders, `DOSTransformer_phonon.py:129-130`)."""
    return nn.Sequential(nn.Linear(n_in, n_hidden), nn.PReLU(), nn.Linear(n_hidden, n_hidden))


def mlp_ln(n_in: int, n_hidden: int) -> nn.Sequential:
    """Linear -> LayerNorm -> PReLU -> Linear (`DOSTransformer_phonon.py:193,203-204`)."""
    return nn.Sequential(nn.Linear(n_in, n_hidden * 2), nn.LayerNorm(n_hidden * 2), nn.PReLU(),
                         nn.Linear(n_hidden * 2, n_hidden))


def _need_gpu(t: torch.Tensor, who: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{who} runs only on an MI355X through libdosx (no CPU fallback): its inputs are on {t.device}")


class _BlockFn(torch.autograd.Function):
    """One autograd node around a (forward program, backward program) pair of ``functional.py``.

    ``fwd(P, *tensors) -> (outputs tuple, ctx)``; ``bwd(P, G, ctx, grads, sink) -> input-gradient tuple``.  Parameter
    gradients are collected in ``G`` (zero-initialised buffers keyed like P) and handed to autograd."""

    @staticmethod
    def forward(ctx, fwd, bwd, names, n_in, *args):
        ins, params = args[:n_in], args[n_in:]
        P = Fn.pack_params({n: p.detach() for n, p in zip(names, params)})
        with torch.no_grad():
            outs, saved = fwd(P, *[t.detach() if torch.is_tensor(t) else t for t in ins])
        ctx.bwd, ctx.P, ctx.names, ctx.saved, ctx.n_in = bwd, P, names, saved, n_in
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        dev = next(iter(ctx.P.values())).device
        G = {k: torch.zeros_like(v) for k, v in ctx.P.items()}
        sink = ops.GradSink(dev)
        with torch.no_grad():
            din = ctx.bwd(ctx.P, G, ctx.saved, [None if g is None else g.float().contiguous() for g in grads], sink)
        sink.flush()
        sink.release()
        return (None, None, None, None) + tuple(din) + tuple(G[n] for n in ctx.names)


def _run(module: nn.Module, prefix: str, fwd, bwd, inputs, dead=()):
    live = [(n, p) for n, p in module.named_parameters() if not any(d in n for d in dead)]
    names = tuple(prefix + n for n, _ in live)
    return _BlockFn.apply(fwd, bwd, names, len(inputs), *inputs, *[p for _, p in live])


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class Encoder(nn.Module):
    """`DOSTransformer_phonon.py:126-145`, `DOSTransformer.py:100-122`, `graphnetwork_phonon.py:132-160`,
    `graphnetwork.py:79-105`.  Creation order matters for same-seed initial values."""

    def __init__(self, n_atom_feats, n_bond_feats, n_hidden, n_global_feats=None, prompt_branch=False):
        super().__init__()
        self.node_encoder = mlp_prelu(n_atom_feats, n_hidden)
        if prompt_branch:
            self.node_encoder_prompt = mlp_prelu(n_atom_feats + n_hidden // 2, n_hidden)
        self.edge_encoder = mlp_prelu(n_bond_feats, n_hidden)
        if n_global_feats is not None:
            self.global_encoder = mlp_prelu(n_global_feats, n_hidden)
        self._n_atom_feats = n_atom_feats

    def forward(self, x, edge_attr, *rest):
        """phonon: ``(x, edge_attr, batch, energies) -> (x, edge_attr, energies)``;
        eDOS: ``(x, edge_attr, glob, batch, energies) -> (x, edge_attr, u, energies)``.  ``energies [S,H]`` is expanded
        to ``[S, B, H]`` with ``B = len(batch.unique())`` like upstream (a view, no arithmetic)."""
        _need_gpu(x, "Encoder")
        has_glob = hasattr(self, "global_encoder")
        if has_glob:
            glob, batch, energies = rest
        else:
            (batch, energies), glob = rest, None
        node_key = "node_encoder"
        if hasattr(self, "node_encoder_prompt") and x.shape[1] != self._n_atom_feats:    # graphnetwork_phonon.py:150-153
            node_key = "node_encoder_prompt"
        keys = [node_key, "edge_encoder"] + (["global_encoder"] if has_glob else [])
        dead = tuple(k + "." for k in ("node_encoder", "node_encoder_prompt", "edge_encoder", "global_encoder") if k not in keys)
        H = self.edge_encoder[2].out_features

        def fwd(P, *ts):
            outs, ctxs = [], []
            for k, t in zip(keys, ts):
                t2 = _f32c(t).reshape(-1, 2) if k == "global_encoder" else _f32c(t)
                y, c = Fn.mlp_prelu_fwd(P, k, Fn.SegList([seg(t2)], [t2]), t2.shape[0], H)
                outs.append(y)
                ctxs.append(c)
            return outs, ctxs

        def bwd(P, G, ctxs, grads, sink):
            for k, c, dy in zip(keys, ctxs, grads):
                if dy is not None:
                    Fn.mlp_prelu_bwd(P, G, k, c, dy, sink)
            return [None] * len(keys)              # raw features carry no gradient upstream either

        ins = [x, edge_attr] + ([glob] if has_glob else [])
        outs = _run(self, "", fwd, bwd, ins, dead)
        nb = int(batch.unique().numel())
        energies = energies.reshape(energies.shape[0], 1, energies.shape[1]).expand(energies.shape[0], nb, energies.shape[1])
        return tuple(outs) + (energies,)


def _local_meta(edge_index: torch.Tensor, n: int, device) -> GraphMeta:
    ei = edge_index.detach().cpu().numpy().astype(np.int64)
    presorted = bool(ei.shape[1] == 0 or np.all(ei[1, 1:] >= ei[1, :-1]))
    return _build_meta_host(ei, np.zeros(n, np.int64), 1, None, presorted).to(device)


class EdgeModel(nn.Module):
    """`DOSTransformer_phonon.py:190-197`: ``edge_mlp(cat[src, dest, edge_attr])``."""

    def __init__(self, n_hidden):
        super().__init__()
        self.edge_mlp = mlp_ln(n_hidden * 3, n_hidden)

    def forward(self, src, dest, edge_attr):
        _need_gpu(src, "EdgeModel")
        H = edge_attr.shape[1]

        def fwd(P, s, d, e):
            s, d, e = _f32c(s), _f32c(d), _f32c(e)
            y, c = Fn.mlp_ln_fwd(P, "edge_mlp", Fn.SegList([seg(s), seg(d), seg(e)], [s, d, e]), e.shape[0], H)
            return [y], c

        def bwd(P, G, c, grads, sink):
            dcat = Fn.mlp_ln_bwd(P, G, "edge_mlp", c, grads[0], sink)
            return [dcat[:, :H].contiguous(), dcat[:, H:2 * H].contiguous(), dcat[:, 2 * H:].contiguous()]

        return _run(self, "", fwd, bwd, [src, dest, edge_attr])[0]


class NodeModel(nn.Module):
    """`DOSTransformer_phonon.py:200-212` (scatter_mean) / `DOSTransformer.py:178-190` (scatter_sum):
    ``node_mlp_2(cat[x, aggregate(edge_attr, col)])``; ``node_mlp_1`` exists and is never used, like upstream."""

    def __init__(self, n_hidden, aggr: str = "mean"):
        super().__init__()
        self.node_mlp_1 = mlp_ln(n_hidden * 2, n_hidden)     # never used upstream either -> grad stays None
        self.node_mlp_2 = mlp_ln(n_hidden * 2, n_hidden)
        self.aggr = aggr

    def forward(self, x, edge_index, edge_attr):
        _need_gpu(x, "NodeModel")
        N, H = x.shape
        m = _local_meta(edge_index, N, x.device)
        mean = self.aggr == "mean"

        def fwd(P, xx, _edge_index, ee):
            xx, ee = _f32c(xx), _f32c(ee)
            if m.edge_perm is not None:
                ee = ee[m.edge_perm].contiguous()
            agg = ops.alloc(xx.device, N, H)
            ops.segment_reduce(ee, m.rowptr_dst, m.inv_deg if mean else None, agg, None, None, N, m.num_edges, H)
            y, c = Fn.mlp_ln_fwd(P, "node_mlp_2", Fn.SegList([seg(xx), seg(agg)], [xx, agg], plain=(xx, agg)), N, H)
            return [y], c

        def bwd(P, G, c, grads, sink):
            dcat = Fn.mlp_ln_bwd(P, G, "node_mlp_2", c, grads[0], sink)                 # [N, 2H]
            de = ops.alloc(dcat.device, m.num_edges, H)
            ops.edge_grad_combine(None, dcat.data_ptr() + 4 * H, 2 * H, m.dst, m.inv_deg if mean else None, de, m.num_edges, H)
            sink._keep.append(dcat)
            if m.edge_perm is not None:
                out = torch.empty_like(de)
                out[m.edge_perm] = de
                de = out
            return [dcat[:, :H].contiguous(), None, de]

        return _run(self, "", fwd, bwd, [x, edge_index, edge_attr], dead=("node_mlp_1.",))[0]


class Processor(nn.Module):
    """`DOSTransformer_phonon.py:148-171`: ``e' = edge_model(x[row], x[col], e)``; ``x' = node_model(x, edge_index, e')``;
    returns ``(x', e')`` (the residuals are the caller's, `:81-84`).  One fused program: gathered K-segments, CSR segment
    reduction, both MLPs — the layer the full models run L times."""

    def __init__(self, edge_model=None, node_model=None):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model

    def forward(self, x, edge_index, edge_attr):
        _need_gpu(x, "Processor")
        if self.edge_model is None or self.node_model is None:
            e2 = edge_attr if self.edge_model is None else self.edge_model(x[edge_index[0]], x[edge_index[1]], edge_attr)
            x2 = x if self.node_model is None else self.node_model(x, edge_index, e2)
            return x2, e2
        N, H = x.shape
        m = _local_meta(edge_index, N, x.device)
        E = m.num_edges
        mean = getattr(self.node_model, "aggr", "mean") == "mean"
        scale = m.inv_deg if mean else None

        def fwd(P, xx, _edge_index, ee):
            xx, ee = _f32c(xx), _f32c(ee)
            if m.edge_perm is not None:
                ee = ee[m.edge_perm].contiguous()
            a_e = Fn.SegList([seg(xx, rmap=ops.rowmap(idx=m.src)), seg(xx, rmap=ops.rowmap(idx=m.dst)), seg(ee)], [xx, ee])
            msg, cxe = Fn.mlp_ln_fwd(P, "edge_model.edge_mlp", a_e, E, H)
            agg = ops.alloc(xx.device, N, H)
            ops.segment_reduce(msg, m.rowptr_dst, scale, agg, None, None, N, E, H)
            xn, cxn = Fn.mlp_ln_fwd(P, "node_model.node_mlp_2", Fn.SegList([seg(xx), seg(agg)], [xx, agg], plain=(xx, agg)), N, H)
            e_out = msg
            if m.edge_perm is not None:                      # back to the caller's edge order
                e_out = torch.empty_like(msg)
                e_out[m.edge_perm] = msg
            return [xn, e_out], (cxe, cxn)

        def bwd(P, G, saved, grads, sink):
            cxe, cxn = saved
            dxn, de_out = grads
            dev = xx_dev = next(iter(P.values())).device
            if dxn is None:
                dxn = ops.zeros(dev, N, H)
            dcat_n = Fn.mlp_ln_bwd(P, G, "node_model.node_mlp_2", cxn, dxn, sink)        # [N, 2H]
            if de_out is not None and m.edge_perm is not None:
                de_out = de_out[m.edge_perm].contiguous()
            dmsg = ops.alloc(dev, E, H)
            ops.edge_grad_combine(de_out, dcat_n.data_ptr() + 4 * H, 2 * H, m.dst, scale, dmsg, E, H)
            dcat_e = Fn.mlp_ln_bwd(P, G, "edge_model.edge_mlp", cxe, dmsg, sink)          # [E, 3H]
            dx = ops.alloc(dev, N, H)
            ops.gather_bwd(dcat_e, dcat_n.data_ptr(), 2 * H, None, m.rowptr_dst, m.rowptr_src, m.perm_src, None, dx, None,
                           N, E, H)
            sink._keep.extend([dcat_n, dcat_e])
            de = dcat_e[:, 2 * H:].contiguous()
            if m.edge_perm is not None:
                out = torch.empty_like(de)
                out[m.edge_perm] = de
                de = out
            return [dx, None, de]

        xn, e2 = _run(self, "", fwd, bwd, [x, edge_index, edge_attr], dead=("node_mlp_1.",))
        return xn, e2


class Decoder(nn.Module):
    """`DOSTransformer_phonon.py:174-183` (``mlp(scatter_sum(x, batch))``) / `DOSTransformer.py:151-161`
    (``mlp(cat[glob, scatter_sum(x, batch)])``)."""

    def __init__(self, n_in, n_hidden):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(n_in, n_hidden))

    def forward(self, x, *rest):
        _need_gpu(x, "Decoder")
        glob, batch = (rest if len(rest) == 2 else (None, rest[0]))
        N, H = x.shape
        bv = batch.detach().cpu().numpy().astype(np.int64)
        B = int(bv.max()) + 1 if bv.size else 0
        order = None
        if bv.size and np.any(bv[1:] < bv[:-1]):      # scatter_sum(x, batch) takes any order: pool a sorted copy
            order_np = np.argsort(bv, kind="stable")
            bv = bv[order_np]
            order = torch.from_numpy(order_np).to(x.device)
            x = x.index_select(0, order)              # (autograd scatters the gradient back)
        counts = np.bincount(bv, minlength=B)
        gptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)).to(x.device)
        ngraph = torch.from_numpy(bv.astype(np.int32)).to(x.device)
        Ho = self.mlp[0].out_features

        def fwd(P, xx, *gg):
            xx = _f32c(xx)
            pooled = ops.alloc(xx.device, B, H)
            ops.graph_pool(xx, gptr, pooled.data_ptr(), H, B, H)
            segs = Fn.SegList([seg(pooled)], [pooled])
            if gg:
                u = _f32c(gg[0])
                segs = Fn.SegList([seg(u), seg(pooled)], [u, pooled])
            y = ops.alloc(xx.device, B, Ho)
            ops.gemm(B, Ho, segs.segs, P["mlp.0.weight"], y, bias=P["mlp.0.bias"])
            return [y], segs

        def bwd(P, G, segs, grads, sink):
            dy = grads[0]
            dev = dy.device
            Fn._wgrad_linear(sink, G, "mlp.0.weight", "mlp.0.bias", B, Ho, seg(dy), segs.segs, keep=(dy,))
            K = segs.K
            dcat = ops.alloc(dev, B, K)
            ops.gemm(B, K, [seg(dy)], P["mlp.0.weight"], dcat, w_layout=1)
            dx = ops.alloc(dev, N, H)
            ops.graph_pool_bwd(dcat.data_ptr() + 4 * (K - H), K, ngraph, dx, N, H, False)
            sink._keep.append(dcat)
            return [dx] + ([dcat[:, :K - H].contiguous()] if K > H else [])

        ins = [x] + ([glob] if glob is not None else [])
        return _run(self, "", fwd, bwd, ins)[0]
