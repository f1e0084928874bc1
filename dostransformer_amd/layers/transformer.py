"""Drop-in for the reference's ``layers/transformer.py`` on MI355X: pre-norm encoder whose layers
re-use the ORIGINAL keys/values (`transformer.py:72-73`) and share ``layer_norms[0]`` between q, k
and v (`:131-134`).  Forward/backward run as the libdosx encoder program (``functional.encoder_*``).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from .multihead_attention import MultiheadAttention


def Linear(in_features, out_features, bias=True):
    m = nn.Linear(in_features, out_features, bias)
    nn.init.xavier_uniform_(m.weight)
    if bias:
        nn.init.constant_(m.bias, 0.)
    return m


def LayerNorm(embedding_dim):
    return nn.LayerNorm(embedding_dim)


class TransformerEncoderLayer(nn.Module):
    """Parameter layout of `transformer.py:98-118`; executed by the enclosing TransformerEncoder."""

    def __init__(self, embed_dim, num_heads=4, attn_dropout=0.0, relu_dropout=0.0, res_dropout=0.0):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.self_attn = MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads, attn_dropout=attn_dropout)
        self.relu_dropout = relu_dropout
        self.res_dropout = res_dropout
        self.normalize_before = True
        self.fc1 = Linear(embed_dim, 4 * embed_dim)
        self.fc2 = Linear(4 * embed_dim, embed_dim)
        self.layer_norms = nn.ModuleList([LayerNorm(embed_dim) for _ in range(2)])

    def forward(self, x, x_k=None, x_v=None, mask=None):
        enc = _single_layer_view(self)
        return enc(x, x_k, x_v, mask, _skip_final_ln=True)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kv, enc, names, skip_final, *params):
        sq, b, h = x.shape
        self_attn = kv is None
        dev = x.device
        P = Fn.pack_params({"enc." + n: p.detach() for n, p in zip(names, params)})
        x2 = x.detach().contiguous().reshape(sq * b, h)
        src = x2 if self_attn else kv.detach().contiguous().reshape(-1, h)
        nk = sq if self_attn else kv.shape[0]
        kvhat = torch.empty(nk * b, h, device=dev)
        rstd = torch.empty(nk * b, device=dev)
        ops.rownorm(src, kvhat, rstd, nk * b, h)
        y, c = Fn.encoder_fwd(P, "enc", x2, sq, b, b, 1, kvhat, nk, b, h, len(enc.layers), final_ln=not skip_final,
                              drop=enc._dropout(dev), fdrop=enc._fdropout(dev))
        enc._bump_seed()
        ctx.c, ctx.P, ctx.names, ctx.params = c, P, names, params
        ctx.kv = (kvhat, rstd, nk, self_attn)
        ctx.dims = (sq, b, h)
        return y.reshape(sq, b, h)

    @staticmethod
    def backward(ctx, dy):
        sq, b, h = ctx.dims
        kvhat, rstd, nk, self_attn = ctx.kv
        dev = dy.device
        G = {k: torch.zeros_like(v) for k, v in ctx.P.items() if ".self_attn." not in k}
        sink = ops.GradSink(dev)
        dkv = torch.zeros(nk * b, h, device=dev)
        dx = Fn.encoder_bwd(ctx.P, G, "enc", ctx.c, dy.contiguous().reshape(sq * b, h).float(), dkv, sink)
        sink.flush()         # (joins the side stream: dkv is complete)
        sink.release()
        dkv_in = None
        if self_attn:
            ops.rownorm_bwd(dkv, kvhat, rstd, dx, sq * b, h, True)
        else:
            dkv_in = torch.empty(nk * b, h, device=dev)
            ops.rownorm_bwd(dkv, kvhat, rstd, dkv_in, nk * b, h, False)
            dkv_in = dkv_in.reshape(nk, b, h)
        grads = tuple(G.get("enc." + n) for n in ctx.names)
        return (dx.reshape(sq, b, h), dkv_in, None, None, None) + grads


class _EncoderKVFn(torch.autograd.Function):
    """The encoder with K != V (x_in_k is not x_in_v and / or embed dropout): functional.encoder_kv_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, xk, xv, enc, names, skip_final, *params):
        sq, b, h = x.shape
        nk = xk.shape[0]
        dev = x.device
        P = {"enc." + n: p.detach() for n, p in zip(names, params)}
        x2 = x.detach().contiguous().reshape(sq * b, h)
        k2 = xk.detach().contiguous().reshape(nk * b, h)
        v2 = xv.detach().contiguous().reshape(nk * b, h)
        emasks = (None, None, None)
        pe = float(enc.dropout or 0.0)
        if enc.training and pe > 0.0:                       # transformer.py:61-68: three independent draws
            seed = enc._seed(dev)
            ms = []
            for i, (t, name) in enumerate(((x2, "x"), (k2, "k"), (v2, "v"))):
                m = torch.empty_like(t)
                ops.dropout_mask(m, pe, seed, 2000 + i)
                if Fn.EDROP_MASK_LOG is not None:
                    Fn.EDROP_MASK_LOG.append(("enc", name, m))
                ms.append(m)
            emasks = tuple(ms)
            xd, kd, vd = torch.empty_like(x2), torch.empty_like(k2), torch.empty_like(v2)
            ops.mask_residual(x2, emasks[0], None, xd, None, sq * b, h)
            ops.mask_residual(k2, emasks[1], None, kd, None, nk * b, h)
            ops.mask_residual(v2, emasks[2], None, vd, None, nk * b, h)
            x2, k2, v2 = xd, kd, vd
        y, c = Fn.encoder_kv_fwd(P, "enc", x2, k2, v2, sq, b, nk, b, h, len(enc.layers), final_ln=not skip_final,
                                 drop=enc._dropout(dev), fdrop=enc._fdropout(dev))
        enc._bump_seed()
        ctx.c, ctx.P, ctx.names, ctx.emasks = c, P, names, emasks
        ctx.dims = (sq, b, nk, h)
        return y.reshape(sq, b, h)

    @staticmethod
    def backward(ctx, dy):
        sq, b, nk, h = ctx.dims
        dev = dy.device
        G = {k: torch.zeros_like(v) for k, v in ctx.P.items() if ".self_attn." not in k}
        sink = ops.GradSink(dev)
        dx, dk, dv = Fn.encoder_kv_bwd(ctx.P, G, "enc", ctx.c, dy.contiguous().reshape(sq * b, h).float(), sink)
        sink.flush()
        sink.release()
        outs = []
        for g_, m, n in ((dx, ctx.emasks[0], sq * b), (dk, ctx.emasks[1], nk * b), (dv, ctx.emasks[2], nk * b)):
            if m is not None:
                o = torch.empty_like(g_)
                ops.mask_residual(g_, m, None, o, None, n, h)
                g_ = o
            outs.append(g_)
        grads = tuple(G.get("enc." + n) for n in ctx.names)
        return (outs[0].reshape(sq, b, h), outs[1].reshape(nk, b, h), outs[2].reshape(nk, b, h), None, None, None) + grads


class TransformerEncoder(nn.Module):
    """`transformer.py:8-44` constructor, `:46-79` forward contract ((seq, batch, dim) tensors)."""

    def __init__(self, embed_dim, num_heads, layers, attn_dropout=0.0, relu_dropout=0.0, res_dropout=0.0,
                 embed_dropout=0.0, attn_mask=False):
        super().__init__()
        self.dropout = embed_dropout
        self.attn_dropout = attn_dropout
        self.embed_dim = embed_dim
        self.embed_scale = math.sqrt(embed_dim)
        self.attn_mask = attn_mask
        self.layers = nn.ModuleList([])
        for _ in range(layers):
            self.layers.append(TransformerEncoderLayer(embed_dim, num_heads=num_heads, attn_dropout=attn_dropout,
                                                       relu_dropout=relu_dropout, res_dropout=res_dropout))
        self.register_buffer('version', torch.Tensor([2]))
        self.normalize = True
        if self.normalize:
            self.layer_norm = LayerNorm(embed_dim)

    def _seed(self, device):
        """device-resident dropout seed (starts from torch's RNG; bumped once per forward call)"""
        seed = getattr(self, "_drop_seed", None)
        if seed is None or seed.device != device:
            seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
            object.__setattr__(self, "_drop_seed", seed)
        return seed

    def _fdropout(self, device):
        """None or (p_relu, p_res, seed_dev, stream_base) for Fn.encoder_fwd: relu / res dropout of the layers (all layers
        of an encoder share the values, like upstream's constructor)."""
        if not self.training or len(self.layers) == 0:
            return None
        p_relu, p_res = float(self.layers[0].relu_dropout or 0.0), float(self.layers[0].res_dropout or 0.0)
        if p_relu <= 0.0 and p_res <= 0.0:
            return None
        return p_relu, p_res, self._seed(device), 1000

    def _dropout(self, device):
        """None (eval / p = 0) or (p, seed_dev, stream_base) for Fn.encoder_fwd; the seed starts from torch's RNG and is
        bumped once per forward call."""
        p = float(self.attn_dropout or 0.0)
        if not self.training or p <= 0.0:
            return None
        return p, self._seed(device), 0

    def _bump_seed(self):
        """one bump per forward call (after the call's masks have been drawn from the current value)"""
        seed = getattr(self, "_drop_seed", None)
        if seed is not None and self.training:
            seed.add_(1)

    def forward(self, x_in, x_in_k=None, x_in_v=None, mask=None, _skip_final_ln=False):
        if x_in_k is None or x_in_v is None:
            # upstream leaves x_k unbound here and crashes (transformer.py:64-73); be explicit instead
            raise ValueError("TransformerEncoder needs x_in_k and x_in_v (pass x_in for self attention)")
        if not x_in.is_cuda:
            raise RuntimeError("TransformerEncoder runs only on an MI355X through libdosx (no CPU fallback)")
        live = [(n, p) for n, p in self.named_parameters() if ".self_attn." not in n]
        names = tuple(n for n, _ in live)
        params = tuple(p for _, p in live)
        if x_in_k is not x_in_v or (self.training and float(self.dropout or 0.0) > 0.0):
            # K != V - different tensors, or embed dropout, which draws different masks for keys and values
            # (transformer.py:61-68): the general, unfused path.  Every reference call site passes one tensor for both and
            # leaves embed dropout at 0 (DOSTransformer_phonon.py:27-38,88,97,99): that is the fused path below.
            y = _EncoderKVFn.apply(x_in.float(), x_in_k.float(), x_in_v.float(), self, names, _skip_final_ln, *params)
            return y.to(x_in.dtype)
        kv = None if x_in_k is x_in else x_in_k.float()
        y = _EncoderFn.apply(x_in.float(), kv, self, names, _skip_final_ln, *params)
        return y.to(x_in.dtype)

    def max_positions(self):
        raise AttributeError("max_positions() is dead code upstream (embed_positions is never defined)")


def _single_layer_view(layer: TransformerEncoderLayer) -> TransformerEncoder:
    enc = TransformerEncoder.__new__(TransformerEncoder)
    nn.Module.__init__(enc)
    enc.dropout, enc.attn_dropout = 0.0, layer.self_attn.attn_dropout
    enc.embed_dim = layer.embed_dim
    enc.layers = nn.ModuleList([layer])
    enc.normalize = False
    enc.training = layer.training
    return enc
