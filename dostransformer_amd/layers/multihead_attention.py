"""Drop-in for the reference's ``layers/multihead_attention.py`` on MI355X.

The reference forward (`multihead_attention.py:49-76`) is, literally,
``softmax_fp32(Q K^T * embed_dim**-0.5) V`` per batch element: the q/k/v/out projections are
commented out upstream (`:63-66,73`), ``attn_mask`` is ignored, ``num_heads`` is unused.  The
projection parameters still exist (state_dict parity) and never receive a gradient.
"""
from __future__ import annotations

import torch
from torch import nn
from torch.nn import Parameter

from .. import ops
from .._lib import Attn

RAW_Q, NO_RESIDUAL = 1, 2


class _BareAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, kv, mask=None):
        sq, b, h = q.shape
        nk = kv.shape[0]
        dev = q.device
        q2 = q.contiguous().reshape(sq * b, h)
        kv2 = kv.contiguous().reshape(nk * b, h)
        ones = torch.ones(h, device=dev)
        zeros = torch.zeros(h, device=dev)
        out = torch.empty(sq * b, h, device=dev)
        probs = torch.empty(b, sq, nk, device=dev)
        a = Attn()
        a.Sq, a.Bq, a.Nk, a.Bk, a.H = sq, b, nk, b, h
        a.q_stride_s, a.q_stride_b, a.flags = b, 1, RAW_Q | NO_RESIDUAL
        a.x, a.kvhat, a.gamma0, a.beta0 = q2.data_ptr(), kv2.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        a.out, a.probs = out.data_ptr(), probs.data_ptr()
        a.drop_mask = mask.data_ptr() if mask is not None else None
        ops.attention_fwd(a)
        ctx.mask = mask
        ctx.save_for_backward(q2, kv2, probs, ones, zeros)
        ctx.dims = (sq, b, nk, h)
        return out.reshape(sq, b, h)

    @staticmethod
    def backward(ctx, dout):
        q2, kv2, probs, ones, zeros = ctx.saved_tensors
        sq, b, nk, h = ctx.dims
        dev = q2.device
        dout2 = dout.contiguous().reshape(sq * b, h).float()
        dq = torch.empty(sq * b, h, device=dev)
        dkv = torch.zeros(nk * b, h, device=dev)
        dsc = torch.empty(b, sq, nk, device=dev)
        nqt, nkt = (sq + 31) // 32, (nk + 31) // 32
        part = torch.empty(b * nqt + b * nkt, 2 * h, device=dev)
        a = Attn()
        a.Sq, a.Bq, a.Nk, a.Bk, a.H = sq, b, nk, b, h
        a.q_stride_s, a.q_stride_b, a.flags = b, 1, RAW_Q | NO_RESIDUAL
        a.x, a.kvhat, a.gamma0, a.beta0 = q2.data_ptr(), kv2.data_ptr(), ones.data_ptr(), zeros.data_ptr()
        a.probs = probs.data_ptr()
        a.dout, a.dx, a.dscores, a.dkvhat, a.dkv_accumulate = dout2.data_ptr(), dq.data_ptr(), dsc.data_ptr(), \
            dkv.data_ptr(), 1
        a.partials_q = part.data_ptr()
        a.partials_kv = part.data_ptr() + 4 * b * nqt * 2 * h
        a.drop_mask = ctx.mask.data_ptr() if ctx.mask is not None else None
        ops.attention_bwd(a)
        return dq.reshape(sq, b, h), dkv.reshape(nk, b, h), None


class _BareAttentionKV(torch.autograd.Function):
    """softmax(Q K^T * dim**-0.5) V in general form - K is not V, or an embedding wider than 256: the
    softmax weights from the MFMA kernel where the shape fits it (else scores + row softmax), the products with V and the
    whole backward from the plain building blocks of csrc/attention_kv.hip."""

    @staticmethod
    def forward(ctx, q, k, v, mask=None):
        sq, b, h = q.shape
        nk = k.shape[0]
        dev = q.device
        q2, k2, v2 = q.contiguous().reshape(sq * b, h), k.contiguous().reshape(nk * b, h), v.contiguous().reshape(nk * b, h)
        probs = torch.empty(b, sq, nk, device=dev)
        ops.attention_weights(q2, k2, probs, sq, b, nk, b, h)  # (MFMA kernel where the shape fits it, general form otherwise)
        out = torch.empty(sq * b, h, device=dev)
        ops.attn_pv(probs, mask, v2, out, sq, b, nk, b, h)
        ctx.mask = mask
        ctx.save_for_backward(q2, k2, v2, probs)
        ctx.dims = (sq, b, nk, h)
        return out.reshape(sq, b, h)

    @staticmethod
    def backward(ctx, dout):
        q2, k2, v2, probs = ctx.saved_tensors
        sq, b, nk, h = ctx.dims
        dev, mask = q2.device, ctx.mask
        dout2 = dout.contiguous().reshape(sq * b, h).float()
        dv = torch.empty(nk * b, h, device=dev)
        ops.attn_tv(probs, mask, dout2, dv, sq, b, nk, b, h)
        dpd = torch.empty(b, sq, nk, device=dev)
        ops.attn_dp(dout2, v2, dpd, sq, b, nk, b, h)
        ds = torch.empty(b, sq, nk, device=dev)
        ops.softmax_bwd(probs, mask, dpd, ds, b * sq, nk, h ** -0.5)
        dq, dk = torch.empty(sq * b, h, device=dev), torch.empty(nk * b, h, device=dev)
        ops.attn_pv(ds, None, k2, dq, sq, b, nk, b, h)
        ops.attn_tv(ds, None, q2, dk, sq, b, nk, b, h)
        return dq.reshape(sq, b, h), dk.reshape(nk, b, h), dv.reshape(nk, b, h), None


class MultiheadAttention(nn.Module):
    """Same constructor / parameters / forward contract as `multihead_attention.py:9-76`."""

    def __init__(self, embed_dim, num_heads, attn_dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.attn_dropout = attn_dropout
        self.scaling = self.embed_dim ** -0.5
        self.in_proj_weight = Parameter(torch.Tensor(3 * embed_dim, embed_dim))
        self.register_parameter('in_proj_bias', None)
        if bias:
            self.in_proj_bias = Parameter(torch.Tensor(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        if add_bias_kv:
            self.bias_k = Parameter(torch.Tensor(1, 1, embed_dim))
            self.bias_v = Parameter(torch.Tensor(1, 1, embed_dim))
        else:
            self.bias_k = self.bias_v = None
        self.add_zero_attn = add_zero_attn
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.xavier_uniform_(self.out_proj.weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)
        if self.bias_k is not None:
            nn.init.xavier_normal_(self.bias_k)
        if self.bias_v is not None:
            nn.init.xavier_normal_(self.bias_v)

    def forward(self, query, key, value, attn_mask=None):
        """(Time, Batch, Channel) in and out; ``attn_mask`` accepted and ignored like upstream."""
        tgt_len, bsz, embed_dim = query.size()
        assert embed_dim == self.embed_dim
        assert list(query.size()) == [tgt_len, bsz, embed_dim]
        assert key.size() == value.size()
        # every reference call site passes the same tensor for key and value (DOSTransformer_phonon.py:88,97,99): that is
        # the fused kernel's case; a different value tensor takes the general path (_BareAttentionKV)
        same_kv = key is value or (key.data_ptr() == value.data_ptr() and key.stride() == value.stride())
        if not query.is_cuda:
            raise RuntimeError("MultiheadAttention runs only on an MI355X through libdosx (no CPU fallback)")
        mask = None
        if self.training and self.attn_dropout > 0.0:          # F.dropout(attn_weights, p, training) (`:70`)
            seed = getattr(self, "_drop_seed", None)
            if seed is None or seed.device != query.device:
                seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(query.device)
                object.__setattr__(self, "_drop_seed", seed)
            else:
                seed.add_(1)
            mask = torch.empty(bsz, tgt_len, key.shape[0], device=query.device, dtype=torch.float32)
            ops.dropout_mask(mask, float(self.attn_dropout), seed, 0)
            self.last_drop_mask = mask
        fits = embed_dim <= ops.ATTN_MAX_H                     # (any number of keys; wider rows take the building blocks)
        if same_kv and fits:
            out = _BareAttention.apply(query.float(), key.float(), mask)
        else:
            out = _BareAttentionKV.apply(query.float(), key.float(), value.float(), mask)
        return out.to(query.dtype)
