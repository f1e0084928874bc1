"""Forward / backward programs of the DOSTransformer hot path, expressed as sequences of libdosx
kernel launches.  Python here only sequences launches and owns buffer lifetimes; every flop is in
``csrc/*.hip``.  Parameters (``P``) and their gradient buffers (``G``) are dicts keyed exactly like
the reference ``state_dict()``.

Layout conventions: every ``[seq, batch, H]`` tensor of the reference is a row-major ``[seq*batch, H]``
matrix with row ``r = s*batch + b``; the two prediction branches (global / system) share one batch
of ``2B`` "crystals" (branch-major inside the batch axis: bq = branch*B + b).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import Attn, Seg
from ._lib import load as _lib_load
from .batch import GraphMeta
from .ops import (ACT_LEAKY, ACT_RELU, EPI_LN, EPI_PRELU_BWD, EPI_PRELU_LN_BWD, EPI_PRELU_LN_BWD_SEG, EPI_RELU_MASK, EPI_ROWLN_BWD, EPI_SEGSUM,
                  PRO_LN_PRELU, PRO_PRELU, PRO_ROWLN, GradSink, rowmap, seg)

Params = Dict[str, torch.Tensor]


def _empty(dev, *shape):
    return ops.alloc(dev, *shape)


def _rows32(m: int) -> int:
    return (m + 31) // 32


class SegList:
    """K-segments of a gathered/concatenated operand + the tensors that keep their memory alive."""

    def __init__(self, segs: Sequence[Seg], keep: Sequence[torch.Tensor], plain: Optional[Sequence[torch.Tensor]] = None):
        self.segs = list(segs)
        self.keep = list(keep)
        self.K = sum(s.width for s in segs)
        self.plain = plain          # the segments as whole row-major tensors, when that is all they are (no row maps)
        self.factor = None          # (x, e, graph meta) behind cat[x[row], x[col], e]: the EdgeModel input (gnn_fwd)
        self.aggsum = None          # (S, rowptr, scale, N, R) when the second Linear ran on aggregated rows (mlp_ln_fwd)


def pack_params(P: Params) -> Params:
    """The parameters as views of ONE buffer when torch allocated them far apart: the fused feed-forward and NodeModel
    kernels address the two weight matrices of a layer through one 2 GiB buffer window (dosx_ffn_*, dosx_mlp_ln_*).  The
    models keep all parameters in one flat buffer anyway; a standalone module's parameters are separate allocations that
    drift apart in a long-lived process."""
    ptrs = [t.data_ptr() for t in P.values()]
    if not ptrs or max(ptrs) - min(ptrs) < (1 << 30):
        return P
    flat = torch.empty(sum((t.numel() + 3) // 4 * 4 for t in P.values()), device=next(iter(P.values())).device)
    out, o = {}, 0
    for k, t in P.items():
        v = flat[o:o + t.numel()].view(t.shape)
        v.copy_(t)
        out[k] = v
        o += (t.numel() + 3) // 4 * 4
    return out


# M-splits of the weight gradients that run alone at the end of the step (round 3: 1.2918 -> 1.2841 ms per cfg2 step with 32,
# 1.2845 with 16; 0 = the throughput-oriented count of dosx_wgrad_splits, tools/exp/ab_tail.sh)
_TAIL_SPLITS = int(__import__("os").environ.get("DOSX_WGRAD_TAIL_SPLITS", "32"))


def _wgrad_linear(sink: GradSink, G: Params, wkey: str, bkey: Optional[str], M: int, N: int, dy: Seg,
                  segs: Sequence[Seg], keep=(), tail: bool = False, dst: Optional[torch.Tensor] = None, **pro) -> None:
    """dW (and db) of y = A W^T + b as split slabs + reduce jobs (launched on the sink's side stream;
    ``keep``: the tensor(s) behind ``dy`` that the caller may drop before the side stream has run)."""
    if wkey not in G:
        return
    K = sum(s.width for s in segs)
    ns = ops.wgrad_splits(M, N, K)
    if _TAIL_SPLITS > 0 and tail:
        # (tail: a weight gradient of the LAST group of the step - its dY exists only at the very end of the backward pass
        #  and it runs with the GPU to itself: latency, not throughput, is what counts there)
        ns = max(ns, min(_TAIL_SPLITS, max(M // 128, 1)))
    # finished mode (include/dosx.h: DosxWgrad.dst): the kernel sums the M-splits itself (last arriver of every tile, fixed
    # order) and writes dW / db - no slab reduction launch; `slab` is its private tile-major scratch
    nsc = ops.wgrad_scratch_floats(N, K, ns)
    slab = sink.scratch(nsc) if nsc else None
    slab_b = sink.scratch(ns * ((N + 63) // 64) * 64) if (bkey is not None and ns > 1) else None
    kw = dict(dst=G[wkey] if dst is None else dst, dst_bias=G[bkey] if bkey is not None else None, **pro)
    keep = tuple(keep) + tuple(t for t in pro.values() if isinstance(t, torch.Tensor))
    # Jobs are DESCRIBED here and launched at the sink's next flush, never on the spot: call sites describe a job where its
    # operands' buffers exist, which may be in front of the launch that fills them (the fused final-LayerNorm backward writes dY of
    # the last layer's fc2 job inside the ffn_bwd launch that follows).  Round 6 removed the launch-as-described mode
    # (DOSX_GROUP_WGRAD=0): it had been unsound since those fusions and nothing ran it.
    sink.defer_wgrad(ops.wgrad_desc(M, N, dy, segs, slab, slab_b, ns, **kw), keep)


# ------------------------------------------------------------------------------------------------
# Encoder MLP: Linear -> PReLU -> Linear          (DOSTransformer_phonon.py:129-130,141-142)
# ------------------------------------------------------------------------------------------------
def mlp_prelu_fwd(P: Params, key: str, a: SegList, M: int, H: int, z: Optional[torch.Tensor] = None):
    """z: the first Linear's output if the caller already has it (phonon edge encoder: ops.edge_embed_sh1)."""
    dev = P[key + ".0.weight"].device
    if z is None:
        z = _empty(dev, M, H)
        ops.gemm(M, H, a.segs, P[key + ".0.weight"], z, bias=P[key + ".0.bias"])
    y = _empty(dev, M, H)
    ops.gemm(M, H, [seg(z)], P[key + ".2.weight"], y, pro=PRO_PRELU, pro_alpha=P[key + ".1.weight"],
             bias=P[key + ".2.bias"])
    return y, (a, z, M, H)


def mlp_prelu_bwd(P: Params, G: Params, key: str, ctx, dy: torch.Tensor, sink: GradSink, dy_seg: Optional[Seg] = None,
                  tail: bool = False):
    a, z, M, H = ctx
    dev = z.device
    alpha = P[key + ".1.weight"]
    dys = dy_seg if dy_seg is not None else seg(dy)
    _wgrad_linear(sink, G, key + ".2.weight", key + ".2.bias", M, H, dys, [seg(z)], keep=(dy,) if dy is not None else (),
                  tail=tail, pro=PRO_PRELU, pro_alpha=alpha)
    rows = ops.gemm_partial_rows(M, H, EPI_PRELU_BWD)
    part = sink.scratch(rows, 1)
    dz = _empty(dev, M, H)
    ops.gemm(M, H, [dys], P[key + ".2.weight"], dz, w_layout=1, epi=EPI_PRELU_BWD, aux=z, epi_alpha=alpha,
             partials=part, partial_ld=1)
    sink.add(part, 0, G[key + ".1.weight"], rows, 1, 1)
    _wgrad_linear(sink, G, key + ".0.weight", key + ".0.bias", M, H, seg(dz), a.segs, keep=(dz,), tail=tail)


_ENC_BWD_PAIR = __import__("os").environ.get("DOSX_ENC_BWD_PAIR", "1") == "1"
# the next layer's node products inside the NodeModel launch (DosxMlpLn.w3): measured 1.1014 vs 1.1018 ms at cfg2, 6.978 vs 6.963 ms
# Electron-DOS (three / two interleaved pairs) - the launch it removes costs what the longer kernel adds; off
_PQ_IN_NODE_MLP = __import__("os").environ.get("DOSX_PQ_IN_NODE_MLP", "0") == "1"
_PQ_IN_NODE_MLP_CS = __import__("os").environ.get("DOSX_PQ_IN_NODE_MLP_CS", "1") == "1"     # ... inside the column-split NodeModel launch


def mlp_prelu_bwd_pair(P: Params, G: Params, first, second, sink: GradSink, tail: bool = False):
    """The backward of TWO encoder MLPs (first / second = (key, ctx, dy)) whose output gradients exist at the same time - the
    node and the edge encoder at the tail of the step: their two input-gradient products as ONE launch (dosx_gemm_pair: each at
    its own tile height in one grid), the same weight-gradient jobs as mlp_prelu_bwd twice."""
    descs, later = [], []
    for key, ctx, dy in (first, second):
        a, z, M, H = ctx
        alpha = P[key + ".1.weight"]
        _wgrad_linear(sink, G, key + ".2.weight", key + ".2.bias", M, H, seg(dy), [seg(z)], keep=(dy,), tail=tail,
                      pro=PRO_PRELU, pro_alpha=alpha)
        rows = ops.gemm_partial_rows(M, H, EPI_PRELU_BWD)
        part = sink.scratch(rows, 1)
        dz = _empty(z.device, M, H)
        descs.append(dict(M=M, N=H, segs=[seg(dy)], w=P[key + ".2.weight"], out=dz, w_layout=1, epi=EPI_PRELU_BWD, aux=z,
                          epi_alpha=alpha, partials=part, partial_ld=1))
        later.append((key, a, M, H, dz, part, rows))
    ops.gemm_pair(descs[0], descs[1])
    for key, a, M, H, dz, part, rows in later:
        sink.add(part, 0, G[key + ".1.weight"], rows, 1, 1)
        _wgrad_linear(sink, G, key + ".0.weight", key + ".0.bias", M, H, seg(dz), a.segs, keep=(dz,), tail=tail)


# ------------------------------------------------------------------------------------------------
# Edge / Node MLP: Linear -> LayerNorm -> PReLU -> Linear     (DOSTransformer_phonon.py:193,204)
# ------------------------------------------------------------------------------------------------
def _wide_ln(H: int) -> bool:
    """2H-wide LayerNorm rows beyond dosx_gemm's full-row epilogues (N <= 512): the unfused row-kernel path."""
    return 2 * H > 512


def _factor_edge(E: int, H: int, m=None) -> bool:
    """Whether the EdgeModel's first Linear (forward product, weight gradient, input gradient) is factored into node parts +
    edge part.  Round 4 (gather / add / normalise as a row kernel of its own, three extra launches per layer and direction):
    from 4 GF of the un-factored product - Electron-DOS only.  Round 5: (a) where the one-launch kernels of csrc/edge_mlp.hip take
    the layer (hidden 64 / 128, a batch `m` with node-aligned row tiles) ALWAYS - the factored layer is then fewer launches
    (backward: 3 instead of 5) as well as fewer flops, also at the CPU-reference shape (Phonon-DOS H 64, 8 crystals: 0.573 ->
    0.557 ms per step, tools/exp/r5_ab_small.sh); (b) otherwise (gathered addends / destination sums inside dosx_gemm's
    epilogues) from DOSX_FACTOR_MIN_GF = 0.5 GF."""
    if not _FACTOR_EDGE_WGRAD:
        return False
    if (m is not None and getattr(m, "seg_tile", None) is not None and _FACTOR_FUSED and _EDGE_ONE_LAUNCH and _EDGE_ONE_LAUNCH_BWD
            and _FACTOR_DGRAD and _FACTOR_ONE_LAUNCH_ALWAYS and ops.edge_mlp_supported(H)):
        return True
    return 2.0 * E * (2 * H) * (3 * H) >= _FACTOR_MIN_GF * 1e9


def _factor_fused(m, H: int) -> bool:
    """Round 5: the factored form with its gather / add inside dosx_gemm's epilogues (forward: DosxGemm.add_p / add_q in
    EPI_LN; backward: EPI_PRELU_LN_BWD_SEG on the batch's node-aligned row tiles, N = 2H <= 256)."""
    return _FACTOR_FUSED and not _wide_ln(H)


def _factor_heads(rows: int, H: int) -> bool:
    """Whether the two output heads multiply their per-crystal K-segments (graph, prompt) once per crystal instead of once per
    energy: from DOSX_FACTOR_HEADS_MIN_GF (1) GF of saved products - the Electron-DOS shapes (4.2 / 2.1 GF); the Phonon-DOS
    benchmark shape (0.27 GF) and small-batch inference keep the two launches fewer."""
    return _FACTOR_HEADS and 2.0 * rows * H * (2.5 * H) >= _FACTOR_HEADS_MIN_GF * 1e9


def _factor_last(E: int, H: int) -> bool:
    """Whether the LAST message-passing layer aggregates its activations per node in front of its second Linear (mlp_ln_fwd,
    aggsum): from DOSX_FACTOR_LAST_MIN_GF GF of that Linear's E-row product."""
    return _FACTOR_LAST and not _wide_ln(H) and 2.0 * E * (2 * H) * H >= _FACTOR_LAST_MIN_GF * 1e9


def _mlp_ln_fused(a: SegList, M: int, H: int) -> bool:
    return a.plain is not None and len(a.plain) <= 2 and ops.mlp_ln_supported(M, a.K, 2 * H, H)


def mlp_ln_bwd_fused(a: SegList, M: int, H: int, dy: torch.Tensor) -> bool:
    """Whether mlp_ln_bwd runs the block's backward as ONE launch (csrc/mlp2.hip) for this input and gradient layout."""
    return _mlp_ln_fused(a, M, H) and dy.stride(1) == 1


def mlp_ln_fwd(P: Params, key: str, a: SegList, M: int, H: int, res: Optional[torch.Tensor] = None, segsum=None, aggsum=None,
               pq_next=None):
    """segsum = (seg_tile, rowptr, scale, agg, e_in, e_out): the second Linear aggregates its rows per destination node in
    its epilogue (DosxGemm EPI_SEGSUM): agg = scale * segment sums of the output, e_out = e_in + output (None: skipped);
    the output itself (the messages) is not written and None is returned for it.
    aggsum = (rowptr, scale, agg, N): the aggregation moved IN FRONT of the second Linear (the last message-passing layer of
    a large edge set: nobody reads its per-edge output) - activation rows summed per destination node, then an N-row GEMM."""
    dev = P[key + ".0.weight"].device
    xhat = _empty(dev, M, 2 * H)
    rstd = _empty(dev, M)
    if segsum is None and a.plain is not None and len(a.plain) <= 2 and ops.mlp_ln_fwd_supported(M, a.K, 2 * H, H):
        # a few hundred rows (the NodeModel: one row per atom): both Linear layers in ONE launch, the intermediate in LDS
        y = _empty(dev, M, H)
        # pq_next = (W1 of the NEXT layer's EdgeModel, pq [N, 4H]): that layer's two node products on the finished rows, same launch
        w3, pq3 = pq_next if pq_next is not None else (None, None)
        ops.mlp_ln_fwd(M, a.plain[0], a.plain[1] if len(a.plain) > 1 else None, P[key + ".0.weight"], P[key + ".0.bias"],
                       P[key + ".1.weight"], P[key + ".1.bias"], P[key + ".2.weight"], P[key + ".3.weight"],
                       P[key + ".3.bias"], res, xhat, rstd, y, w3=w3, nb3=2 if w3 is not None else 0, pq=pq3)
        return y, (a, xhat, rstd, M, H)
    fac = a.factor
    if fac is not None and _factor_edge(M, H, fac[2]):
        # Large edge sets (throughput-bound): the first Linear FACTORED - Linear(cat[x[row], x[col], e]) = (x Wa^T)[row] +
        # (x Wb^T)[col] + e Wc^T + b.  The two node products are N-row GEMMs, the E-row GEMM keeps a third of the columns, and one
        # row kernel gathers, adds and normalises: a third of the flops of the gathered-concat GEMM (Electron-DOS: 14.1 -> 5.5 GF
        # per layer) for one extra pass over [E, 2H].  Same saved tensors (xhat, rstd): the backward does not care.
        x, e, m = fac
        W1 = P[key + ".0.weight"]
        N_ = m.num_nodes
        pq = getattr(a, "pq_ready", None)       # (the previous layer's NodeModel launch already multiplied them: DosxMlpLn.w3)
        if pq is None:
            pq = _empty(dev, N_, 4 * H)
            ops.gemm_pair(dict(M=N_, N=2 * H, segs=[seg(x)], w=W1[:, :H], out=pq[:, :2 * H]),
                          dict(M=N_, N=2 * H, segs=[seg(x)], w=W1[:, H:2 * H], out=pq[:, 2 * H:]))
        if _factor_fused(m, H) and _EDGE_ONE_LAUNCH and segsum is not None and ops.edge_mlp_supported(H):
            # the whole EdgeModel + aggregation + edge residual in ONE launch on the node-aligned row tiles (csrc/edge_mlp.hip):
            # the [48, 2H] intermediate stays in LDS, the messages never exist in HBM
            tile, rowptr, scale, agg, e_in, e_out = segsum
            ops.edge_mlp_fwd(M, H, e, pq, m.src, m.dst, W1[:, 2 * H:], P[key + ".0.bias"], P[key + ".1.weight"], P[key + ".1.bias"],
                             P[key + ".2.weight"], P[key + ".3.weight"], P[key + ".3.bias"], xhat, rstd, e_out, tile, rowptr, scale, agg)
            a.keep.append(pq)
            return None, (a, xhat, rstd, M, H)
        if _factor_fused(m, H):
            # the E-row product of K = H with the two gathered node rows added in front of its LayerNorm statistics
            ops.gemm(M, 2 * H, [seg(e)], W1[:, 2 * H:], xhat, bias=P[key + ".0.bias"], epi=EPI_LN, aux_out=rstd,
                     add_p=pq[:, :2 * H], add_ip=m.src, add_q=pq[:, 2 * H:], add_iq=m.dst)
        else:
            z = _empty(dev, M, 2 * H)
            ops.gemm(M, 2 * H, [seg(e)], W1[:, 2 * H:], z, bias=P[key + ".0.bias"])
            ops.gather_add_rownorm(z, pq[:, :2 * H], pq[:, 2 * H:], m.src, m.dst, xhat, rstd, M, 2 * H)
        a.keep.append(pq)
    elif _wide_ln(H):
        # hidden > 256: the 2H-wide LayerNorm row no longer fits the one-tile row epilogue of dosx_gemm - plain GEMM, then
        # the parameter-free normalisation as a row kernel (the affine + PReLU stay in the second GEMM's prologue)
        assert segsum is None
        if 2 * H > 1024:
            from ._lib import DosxError
            raise DosxError(f"hidden <= 512: the LayerNorm prologue of dosx_gemm holds rows of up to 1024 floats, 2 * hidden = {2 * H}")
        z = _empty(dev, M, 2 * H)
        ops.gemm(M, 2 * H, a.segs, P[key + ".0.weight"], z, bias=P[key + ".0.bias"])
        ops.rownorm(z, xhat, rstd, M, 2 * H)
    else:
        ops.gemm(M, 2 * H, a.segs, P[key + ".0.weight"], xhat, bias=P[key + ".0.bias"], epi=EPI_LN, aux_out=rstd)
    if aggsum is not None:
        # scale_n * sum_{e -> n} (act_e W^T + b) = (scale_n * sum act_e) W^T + c_n b: E / N times fewer rows through the GEMM
        rowptr, scale, agg, N_ = aggsum
        S, R = _empty(dev, N_, 2 * H), _empty(dev, N_, H)
        ops.act_segment_sum(xhat, rowptr, scale, P[key + ".1.weight"], P[key + ".1.bias"], P[key + ".2.weight"],
                            P[key + ".3.bias"], S, R, N_, M, 2 * H, H)
        ops.gemm(N_, H, [seg(S)], P[key + ".3.weight"], agg, res=R)
        a.aggsum = (S, rowptr, scale, N_, R)
        return None, (a, xhat, rstd, M, H)
    if segsum is not None:
        tile, rowptr, scale, agg, e_in, e_out = segsum
        ops.gemm(M, H, [seg(xhat)], P[key + ".3.weight"], e_out, pro=PRO_LN_PRELU, pro_gamma=P[key + ".1.weight"],
                 pro_beta=P[key + ".1.bias"], pro_alpha=P[key + ".2.weight"], bias=P[key + ".3.bias"],
                 res=e_in if e_out is not None else None, epi=EPI_SEGSUM, seg_tile=tile, seg_rowptr=rowptr, seg_scale=scale,
                 seg_agg=agg)
        return None, (a, xhat, rstd, M, H)
    y = _empty(dev, M, H)
    ops.gemm(M, H, [seg(xhat)], P[key + ".3.weight"], y, pro=PRO_LN_PRELU, pro_gamma=P[key + ".1.weight"],
             pro_beta=P[key + ".1.bias"], pro_alpha=P[key + ".2.weight"], bias=P[key + ".3.bias"], res=res)
    return y, (a, xhat, rstd, M, H)


def mlp_ln_bwd(P: Params, G: Params, key: str, ctx, dy: torch.Tensor, sink: GradSink, res: Optional[torch.Tensor] = None,
               res_col0: int = 0, add_dy: bool = False, pre: Optional[dict] = None) -> torch.Tensor:
    """Returns dL/d(concatenated input) [M, K_in] (+ ``res`` added to its columns [res_col0, K_in)).
    add_dy (one-launch path only, see mlp_ln_bwd_fused): + dy on the first H columns - the residual connection around the
    block, x' = x + MLP(cat[x, .]), differentiated in the same launch."""
    a, xhat, rstd, M, H = ctx
    dev = xhat.device
    gam, bet, alpha = P[key + ".1.weight"], P[key + ".1.bias"], P[key + ".2.weight"]
    agg_first = a.aggsum
    if agg_first is not None:
        # forward aggregated in front of the second Linear (mlp_ln_fwd, aggsum): dy = dL/d agg, one row per NODE (a strided
        # view of the node-MLP input gradient).  Weight gradient sum_n dagg_n (x) S_n on N rows; bias gradient = column sums of
        # c_n * dagg_n (rides in the flush launch as a row-partial reduction); dL/d act_e = scale_n * (dagg_n W)[dst(e)].
        S, rowptr, scale, N_, R = agg_first
        with ops.graph_rows():
            _wgrad_linear(sink, G, key + ".3.weight", None, N_, H, seg(dy), [seg(S)], keep=(dy, S, R))
        daggc = sink.scratch(N_, H)
        ops.seg_count_scale(dy.data_ptr(), int(dy.stride(0)), rowptr, scale is not None, daggc, N_, H)
        sink.add(daggc, 0, G[key + ".3.bias"], N_, H, H)
        dnode = _empty(dev, N_, 2 * H)
        ops.gemm(N_, 2 * H, [seg(dy)], P[key + ".3.weight"], dnode, w_layout=1)
        sink._keep.append(dnode)
    else:
        _wgrad_linear(sink, G, key + ".3.weight", key + ".3.bias", M, H, seg(dy), [seg(xhat)], keep=(dy,), pro=PRO_LN_PRELU,
                      pro_gamma=gam, pro_beta=bet, pro_alpha=alpha)
    fused = mlp_ln_bwd_fused(a, M, H, dy) and agg_first is None and res is None
    assert fused or not add_dy
    assert pre is None or fused           # (pre: what produces dy, in the same column-split launch - see node_chain_ok)
    wide = _wide_ln(H) or agg_first is not None
    fac_dgrad = a.factor is not None and _factor_edge(M, H, a.factor[2]) and _FACTOR_DGRAD and not fused
    # round 5: the destination-node sums of dz (the factored weight / input gradients below need them) inside the dgrad GEMM's
    # epilogue, on the batch's node-aligned row tiles - one partial row per tile
    seg_bwd = (fac_dgrad and agg_first is None and not wide and _factor_fused(a.factor[2], H) and a.factor[2].seg_tile is not None
               and 2 * H <= 256)
    aggD_epi = None
    if seg_bwd:
        m_ = a.factor[2]
        rows = int(m_.seg_tile.shape[1]) - 1
        aggD_epi = sink.scratch(m_.num_nodes, 2 * H)
    else:
        rows = ops.mlp_ln_bwd_partial_rows(M) if fused else (ops.ln_prelu_bwd_partial_rows(M) if wide else
                                                             ops.gemm_partial_rows(M, 2 * H, EPI_PRELU_LN_BWD))
    pld = 4 * H + 4          # [dgamma(2H) | dbeta(2H) | pad(3) | dalpha]; multiple of 4 -> vector reduce
    part = sink.scratch(rows, pld)
    dz = _empty(dev, M, 2 * H)
    dcat = None if fac_dgrad else _empty(dev, M, a.K)
    if fused:
        ops.mlp_ln_bwd(M, dy, xhat, rstd, P[key + ".0.weight"], P[key + ".3.weight"], gam, bet, alpha, dz, dcat, part, add_dy=add_dy, pre=pre)
    elif agg_first is not None:
        ops.ln_prelu_bwd_gather(dnode, a.factor[2].dst, agg_first[2], xhat, rstd, gam, bet, alpha, dz, part, M, 2 * H)
    elif wide:          # plain dgrad GEMM, then PReLU + LayerNorm backward of the 2H-wide rows as a row kernel
        dact = _empty(dev, M, 2 * H)
        ops.gemm(M, 2 * H, [seg(dy)], P[key + ".3.weight"], dact, w_layout=1)
        ops.ln_prelu_bwd(dact, xhat, rstd, gam, bet, alpha, dz, part, M, 2 * H)
        sink._keep.append(dact)
    elif seg_bwd:
        ops.gemm(M, 2 * H, [seg(dy)], P[key + ".3.weight"], dz, w_layout=1, epi=EPI_PRELU_LN_BWD_SEG, aux=xhat,
                 aux_stats=rstd, epi_gamma=gam, epi_beta=bet, epi_alpha=alpha, partials=part, partial_ld=pld,
                 seg_tile=m_.seg_tile, seg_rowptr=m_.rowptr_dst, seg_agg=aggD_epi)
    else:
        ops.gemm(M, 2 * H, [seg(dy)], P[key + ".3.weight"], dz, w_layout=1, epi=EPI_PRELU_LN_BWD, aux=xhat,
                 aux_stats=rstd, epi_gamma=gam, epi_beta=bet, epi_alpha=alpha, partials=part, partial_ld=pld)
    sink.add(part, 0, G[key + ".1.weight"], rows, pld, 2 * H)
    sink.add(part, 2 * H, G[key + ".1.bias"], rows, pld, 2 * H)
    sink.add(part, pld - 1, G[key + ".2.weight"], rows, pld, 1)
    fac = a.factor
    aggS = aggD = None
    if fac is not None and _factor_edge(M, H, fac[2]) and (key + ".0.weight" in G or fac_dgrad):
        # The first Linear reads cat[x[row], x[col], e] (DOSTransformer_phonon.py:193-195): its weight gradient is
        #   sum_e dz_e (x) [x[row(e)] | x[col(e)] | e_e]  =  [ sum_n S_n (x) x_n | sum_n D_n (x) x_n | sum_e dz_e (x) e_e ],
        # S_n / D_n = the sums of dz over the edges that leave / enter node n.  The two node blocks become N-row jobs (20 x
        # fewer rows at 20 edges per node) behind two memory-bound segment sums that run on the weight-gradient stream in
        # front of the group; the edge block keeps its E rows on a third of the columns.  Three jobs write the three column
        # blocks of the one gradient (DosxWgrad.ldd); fixed summation orders, like everything else here.
        x, e, m = fac
        N_, E_ = m.num_nodes, m.num_edges
        aggS = sink.scratch(N_, 2 * H)
        aggD = aggD_epi if aggD_epi is not None else sink.scratch(N_, 2 * H)

        # (the source sums: by the caller's dosx_node_grad launch when that form runs, gnn_bwd)
        node_one = fac_dgrad and _NODE_GRAD_ONE_LAUNCH and _factor_fused(m, H) and H in (64, 128, 256)

        def node_sums(dz=dz, aggS=aggS, aggD=aggD, m=m, with_dst=aggD_epi is None, with_src=not node_one):
            if with_src:
                ops.segment_reduce_perm(dz, m.rowptr_src, m.perm_src, aggS, N_, E_, 2 * H)
            if with_dst:                               # (else: written by the dgrad GEMM's epilogue above)
                ops.segment_reduce(dz, m.rowptr_dst, None, aggD, None, None, N_, E_, 2 * H)
        if fac_dgrad:
            node_sums()                                # the input gradient below reads them too: on the main stream, now
            sink._keep.append(dz)
        else:
            sink.defer_pre(node_sums, keep=(dz,))
        if key + ".0.weight" in G:
            Gw = G[key + ".0.weight"]                  # [2H, 3H]

            # the source block's job reads aggS.  With node_one that is written LATER, by the caller's dosx_node_grad launch:
            # the job is handed back as a closure and gnn_bwd describes it behind that launch, so that no flush between here and
            # there can launch it early
            def src_job(Gw=Gw, aggS=aggS, x=x):
                with ops.graph_rows():
                    _wgrad_linear(sink, G, key + ".0.weight", None, N_, 2 * H, seg(aggS), [seg(x)], keep=(aggS, x), dst=Gw[:, :H])
            if not node_one:
                src_job()
                src_job = None
            with ops.graph_rows():
                _wgrad_linear(sink, G, key + ".0.weight", None, N_, 2 * H, seg(aggD), [seg(x)], keep=(aggD, x), dst=Gw[:, H:2 * H])
                _wgrad_linear(sink, G, key + ".0.weight", key + ".0.bias", M, 2 * H, seg(dz), [seg(e)], keep=(dz, e), dst=Gw[:, 2 * H:])
        else:
            src_job = None
    else:
        src_job = None
        _wgrad_linear(sink, G, key + ".0.weight", key + ".0.bias", M, 2 * H, seg(dz), a.segs, keep=(dz,))
    if fac_dgrad:
        # ... and the INPUT gradient factored the same way: dL/de = dz Wc (+ the incoming edge-state gradient) is the only
        # E-row product left (a third of the columns of the [E,3H] concat gradient); the node parts come from the node sums,
        # sum_{e: row(e) = n} dz_e Wa = S_n Wa - the caller (gnn_bwd) multiplies N rows instead of gathering E of them back
        W0 = P[key + ".0.weight"]
        de_new = _empty(dev, M, H)
        ops.gemm(M, H, [seg(dz)], W0[:, 2 * H:], de_new, w_layout=1, res=res)
        return ("factored", de_new, aggS, aggD, dz if node_one else None, src_job if node_one else None, False)
    if not fused:
        ops.gemm(M, a.K, [seg(dz)], P[key + ".0.weight"], dcat, w_layout=1, res=res, res_col0=res_col0 if res is not None else 0)
    return dcat


# Round 6: the N-row launch in front of a NodeModel backward - the node side of the later layer's input gradient (dosx_node_grad),
# or the dense-key / pooled-decoder backward in front of the last layer's - runs INSIDE that column-split launch (DosxMlpLnBwd.pre)
# (measured, tools/exp/r6_run2.sh / r6_run3.sh, cfg2 step, three interleaved rounds each: the dense-key launch absorbed 1.0940 ->
#  1.0726 ms - kept; the node side absorbed as well 1.0893 -> 1.1018 ms, with the layer's weight-gradient group flushed behind the
#  launch instead of in front of it 1.1102 ms - three in-launch exchanges under a weight-gradient group cost more than the launch
#  they save: off)
_NODE_CHAIN = __import__("os").environ.get("DOSX_NODE_CHAIN", "0") == "1"
_DENSE_CHAIN = __import__("os").environ.get("DOSX_DENSE_CHAIN", "1") == "1"
# ... and the layer's weight-gradient group is then flushed BEHIND that launch (it runs alone, all CUs) instead of in front of it
# the early gradient bucket's flush (transformers / heads: `mid_hook`) BEHIND the last layer's NodeModel backward launch instead of
# in front of it - that launch then does not start under a freshly launched weight-gradient group: 1.0726 -> 1.0690 ms per cfg2
# step, three interleaved rounds (tools/exp/r6_run4.sh).  Hidden 256 (cfg3 / its T4 shard, tools/exp/r6_run9.sh): 6.968 -> 6.997
# and 6.875 -> 6.912 ms - the NodeModel backward is a row-tile kernel there and the early group is 4x the work: hidden <= 128 only
_MID_HOOK_LATE = __import__("os").environ.get("DOSX_MID_HOOK_LATE", "1") == "1"
_MID_HOOK_LATE_MAX_H = 128
_EDGE_ENC_ONE_LAUNCH = __import__("os").environ.get("DOSX_EDGE_ENC_ONE_LAUNCH", "1") == "1"
_FFN_MULTI = __import__("os").environ.get("DOSX_FFN_MULTI", "1") == "1"   # an encoder stack's layers in one forward launch
_ENC_CS = __import__("os").environ.get("DOSX_ENC_CS", "1") == "1"        # node encoder + layer 0's node products: one column-split launch


def dev_of(P: Params):
    return next(iter(P.values())).device


_HEADS_BWD_ONE_LAUNCH = __import__("os").environ.get("DOSX_HEADS_BWD_ONE_LAUNCH", "1") == "1"
_HEADS_BWD_MAX_H = int(__import__("os").environ.get("DOSX_HEADS_BWD_MAX_H", "128"))
_FLUSH_AFTER_CHAIN = __import__("os").environ.get("DOSX_FLUSH_AFTER_CHAIN", "0") == "1"


def node_chain_ok(cxn, N: int, H: int) -> bool:
    """Whether the NodeModel backward described by ``cxn`` (mlp_ln_fwd's context) will run as the column-split one-launch kernel
    on a contiguous [N, H] gradient - the form that takes ``pre``."""
    a = cxn[0]
    return (_mlp_ln_fused(a, N, H) and a.aggsum is None and ops.MLP_LN_CS_BWD != "0" and ops.mlp_ln_cs(N, 2 * H, 2 * H, H))


def edge_bwd_one_launch_ok(a: SegList, M: int, H: int) -> bool:
    """Whether the EdgeModel's backward runs as ONE launch (csrc/edge_mlp.hip, edge_bwd_kernel): factored first Linear, the
    batch's node-aligned row tiles, hidden 64 / 128, per-edge second Linear."""
    return (a.factor is not None and a.aggsum is None and _factor_edge(M, H, a.factor[2]) and _FACTOR_DGRAD and _factor_fused(a.factor[2], H)
            and _EDGE_ONE_LAUNCH_BWD and a.factor[2].seg_tile is not None and ops.edge_mlp_supported(H))


def edge_mlp_bwd_one_launch(P: Params, G: Params, key: str, ctx, dagg: torch.Tensor, de_next: Optional[torch.Tensor], scale,
                            sink: GradSink):
    """dagg [N, H] (a strided view): gradient of the aggregate; de_next: gradient of e_{l+1} or None.  Returns what
    mlp_ln_bwd's factored branch returns: ("factored", dL/de_l [E,H], aggS, aggD)."""
    a, xhat, rstd, M, H = ctx
    x, e, m = a.factor
    dev = xhat.device
    N_, E_ = m.num_nodes, m.num_edges
    gam, bet, alpha = P[key + ".1.weight"], P[key + ".1.bias"], P[key + ".2.weight"]
    W0 = P[key + ".0.weight"]
    rows = int(m.seg_tile.shape[1]) - 1
    pld = 4 * H + 4
    part = sink.scratch(rows, pld)
    dmsg, dz, de_new = _empty(dev, M, H), _empty(dev, M, 2 * H), _empty(dev, M, H)
    aggS, aggD = sink.scratch(N_, 2 * H), sink.scratch(N_, 2 * H)
    ops.edge_mlp_bwd(M, H, dagg, de_next, m.dst, xhat, rstd, P[key + ".3.weight"], W0[:, 2 * H:], gam, bet, alpha, dmsg, dz, de_new,
                     part, m.seg_tile, m.rowptr_dst, scale, aggD)
    _wgrad_linear(sink, G, key + ".3.weight", key + ".3.bias", M, H, seg(dmsg), [seg(xhat)], keep=(dmsg,), pro=PRO_LN_PRELU,
                  pro_gamma=gam, pro_beta=bet, pro_alpha=alpha)
    sink.add(part, 0, G[key + ".1.weight"], rows, pld, 2 * H)
    sink.add(part, 2 * H, G[key + ".1.bias"], rows, pld, 2 * H)
    sink.add(part, pld - 1, G[key + ".2.weight"], rows, pld, 1)
    node_one = _NODE_GRAD_ONE_LAUNCH          # the source sums are then made by the caller's dosx_node_grad launch (gnn_bwd)
    if not node_one:
        ops.segment_reduce_perm(dz, m.rowptr_src, m.perm_src, aggS, N_, E_, 2 * H)
    sink._keep.extend(t for t in (dz, dagg, de_next) if t is not None)
    src_job = None
    if key + ".0.weight" in G:
        Gw = G[key + ".0.weight"]

        def src_job(Gw=Gw, aggS=aggS, x=x):           # the source block's job reads aggS: described once that exists (see gnn_bwd)
            with ops.graph_rows():
                _wgrad_linear(sink, G, key + ".0.weight", None, N_, 2 * H, seg(aggS), [seg(x)], keep=(aggS, x), dst=Gw[:, :H])
        if not node_one:             # (aggS exists: made by the segment_reduce_perm launch above)
            src_job()
            src_job = None
        with ops.graph_rows():
            _wgrad_linear(sink, G, key + ".0.weight", None, N_, 2 * H, seg(aggD), [seg(x)], keep=(aggD, x), dst=Gw[:, H:2 * H])
            _wgrad_linear(sink, G, key + ".0.weight", key + ".0.bias", M, 2 * H, seg(dz), [seg(e)], keep=(dz, e), dst=Gw[:, 2 * H:])
    # last field: whether the layer's weight-gradient group may be flushed IN FRONT of the caller's node-side launch (gnn_bwd)
    return ("factored", de_new, aggS, aggD, dz if node_one else None, src_job, node_one and _GNN_FLUSH_BEFORE_NODE)


# ------------------------------------------------------------------------------------------------
# Message passing stack       (DOSTransformer_phonon.py:81-84,148-171 / DOSTransformer.py:56-59)
# ------------------------------------------------------------------------------------------------
def gnn_fwd(P: Params, m: GraphMeta, x: torch.Tensor, e: torch.Tensor, L: int, mean: bool, H: int, pq0=None):
    """pq0: layer 0's node products [N, 4H] when the node encoder's launch already multiplied them (ops.enc_cs_fwd)."""
    N, E = m.num_nodes, m.num_edges
    dev = x.device
    scale = m.inv_deg if mean else None
    ctxs = []
    pq_ready = pq0
    for l in range(L):
        pre = f"stacked_processor.{l}"
        a_e = SegList([seg(x, rmap=rowmap(idx=m.src)), seg(x, rmap=rowmap(idx=m.dst)), seg(e)], [x, e])
        a_e.factor = (x, e, m)                  # (the parts behind the gathered concat: mlp_ln_bwd factors the weight gradient)
        a_e.pq_ready = pq_ready
        pq_ready = None
        agg = _empty(dev, N, H)
        last = l == L - 1                       # the last layer's edge update is dead
        e_new = None if last else _empty(dev, E, H)
        if last and _factor_last(E, H):
            _, cxe = mlp_ln_fwd(P, pre + ".edge_model.edge_mlp", a_e, E, H, aggsum=(m.rowptr_dst, scale, agg, N))
        elif m.seg_tile is not None and H <= 256:
            # scatter_mean / scatter_sum + the edge residual inside the message GEMM's epilogue (node-aligned row tiles):
            # the messages never reach HBM and the layer is 4 launches instead of 5
            _, cxe = mlp_ln_fwd(P, pre + ".edge_model.edge_mlp", a_e, E, H, segsum=(m.seg_tile, m.rowptr_dst, scale, agg, e, e_new))
        else:
            msg, cxe = mlp_ln_fwd(P, pre + ".edge_model.edge_mlp", a_e, E, H)
            ops.segment_reduce(msg, m.rowptr_dst, scale, agg, e, e_new, N, E, H)
        a_n = SegList([seg(x), seg(agg)], [x, agg], plain=(x, agg))
        pq_next = None
        # (round 6: with the COLUMN-SPLIT NodeModel launch - an eighth of the extra weights per workgroup, one more in-launch
        #  exchange - it pays: always there; the one-workgroup-per-tile form keeps the launch, DOSX_PQ_IN_NODE_MLP)
        cs_node = ops.mlp_ln_cs(N, 2 * H, 2 * H, H) and ops.mlp_ln_fwd_supported(N, 2 * H, 2 * H, H) and _PQ_IN_NODE_MLP_CS
        if (l + 1 < L and _factor_edge(E, H, m) and ops.mlp_ln_fwd_supported(N, 2 * H, 2 * H, H) and
                (cs_node or (_PQ_IN_NODE_MLP and (2 * H) % 256 == 0))):
            # the next layer's node products (x_new Wa^T | x_new Wb^T) ride in this NodeModel launch: one launch fewer per layer
            pq_ready = _empty(dev, N, 4 * H)
            pq_next = (P[f"stacked_processor.{l + 1}.edge_model.edge_mlp.0.weight"], pq_ready)
        x_new, cxn = mlp_ln_fwd(P, pre + ".node_model.node_mlp_2", a_n, N, H, res=x, pq_next=pq_next)
        ctxs.append((cxe, cxn))
        x, e = x_new, e_new
    return x, ctxs


_SPLIT_LATE_FLUSH = int(__import__("os").environ.get("DOSX_SPLIT_LATE_FLUSH", "1"))


def gnn_bwd(P: Params, G: Params, m: GraphMeta, ctxs, dx: torch.Tensor, sink: GradSink, L: int, mean: bool, H: int, dx_pre=None):
    """Returns (dL/dx_0 [N,H], dL/de_0 as a strided [E,H] view or None).
    dx_pre: None, or (pre, launch): ``dx`` has NOT been computed yet - ``pre`` is the ops.mlp_ln_bwd ``pre`` dictionary that makes
    it inside the last layer's NodeModel backward launch, ``launch()`` the stand-alone launch for when that form does not apply."""
    N, E = m.num_nodes, m.num_edges
    dev = dx.device
    pending = None          # the `pre` of the NEXT NodeModel backward: this layer's node-side launch, deferred into it
    if dx_pre is not None:
        if _DENSE_CHAIN and node_chain_ok(ctxs[L - 1][1], N, H) and dx.is_contiguous():
            pending = dx_pre[0]
        else:
            dx_pre[1]()
    scale = m.inv_deg if mean else None
    de = None           # dL/de_{l+1}: the e-block (columns [2H,3H)) of the next layer's concat gradient

    def flush_side(l):
        """sink.flush_on_side(); a flush while layer 0 is being processed launches the last pending job of layers L-1 .. 1:
        from there on their gradients are final - the data-parallel step starts their all-reduce (sink.gnn_hook, train.Trainer)"""
        sink.flush_on_side()
        hook = getattr(sink, "gnn_hook", None)
        if l == 0 and hook is not None:
            sink.gnn_hook = None
            hook(sink)
    for l in reversed(range(L)):
        pre = f"stacked_processor.{l}"
        cxe, cxn = ctxs[l]
        # factored edge layer + one-launch NodeModel backward: the residual path's dx rides on the first H columns of dcat_n
        fold_dx = (cxe[0].factor is not None and _factor_edge(E, H, m) and _FACTOR_DGRAD and _factor_fused(m, H)
                   and mlp_ln_bwd_fused(cxn[0], N, H, dx))
        dcat_n = mlp_ln_bwd(P, G, pre + ".node_model.node_mlp_2", cxn, dx, sink, add_dy=fold_dx, pre=pending)          # [N, 2H]
        if pending is not None and pending.get("flush_after"):
            flush_side(l)
        pending = None
        hook1 = getattr(sink, "after_first_node", None)
        if hook1 is not None:
            sink.after_first_node = None
            hook1(sink)
        if edge_bwd_one_launch_ok(cxe[0], E, H):
            # gather + add of the message gradient, both input-gradient products with the PReLU / LayerNorm backward between
            # them, the destination-node sums: one launch on the node-aligned row tiles (csrc/edge_mlp.hip)
            dcat_e = edge_mlp_bwd_one_launch(P, G, pre + ".edge_model.edge_mlp", cxe, dcat_n[:, H:], de, scale, sink)
        else:
            if cxe[0].aggsum is not None:
                assert de is None                        # (the last layer: no edge-state gradient arrives)
                dmsg = dcat_n[:, H:]                     # dL/d agg [N,H]: mlp_ln_bwd expands it per edge inside its row kernel
            else:
                dmsg = _empty(dev, E, H)
                ops.edge_grad_combine(de, dcat_n.data_ptr() + 4 * H, 2 * H, m.dst, scale, dmsg, E, H)
            # e_{l+1} = e_l + msg_l (DOSTransformer_phonon.py:84): dL/de_l = dL/de_{l+1} + (edge-MLP input gradient)[:, 2H:3H].
            # The dgrad GEMM adds dL/de_{l+1} to exactly those columns, so the e-block of dcat_e IS dL/de_l and no kernel
            # ever streams the edge gradient on its own.
            dcat_e = mlp_ln_bwd(P, G, pre + ".edge_model.edge_mlp", cxe, dmsg, sink, res=de, res_col0=2 * H)   # [E, 3H]
        node_one = isinstance(dcat_e, tuple) and len(dcat_e) > 4 and dcat_e[4] is not None
        if l == 0 and sink.wside is not None and not node_one:
            flush_side(l)     # layer 0's weight gradients start now, under the gather backward and the encoders' backward
        dx_old = _empty(dev, N, H)
        if isinstance(dcat_e, tuple):
            # factored input gradient (large edge sets): dx_l = dx_{l+1} + S Wa + D Wb + (node-MLP input gradient)[:, :H]
            _, de_new, aggS, aggD = dcat_e[:4]
            W0 = P[pre + ".edge_model.edge_mlp.0.weight"]
            if node_one:
                src_job = dcat_e[5]                    # the source block's weight-gradient job: reads the sums the launch below makes
                early = bool(dcat_e[6])                # the layer's group starts BEFORE the node-side launch, under it
                chain = _NODE_CHAIN and l > 0 and node_chain_ok(ctxs[l - 1][1], N, H)
                do_flush = early and sink.wside is not None and (l == 0 or (_SPLIT_LATE_FLUSH == 1 and l == 1) or (_SPLIT_LATE_FLUSH == 2 and l >= 1))
                if do_flush and not (chain and _FLUSH_AFTER_CHAIN):
                    flush_side(l)
                # source sums of dz + both node products + the residual terms: ONE launch (csrc/edge_mlp.hip, node_grad_kernel) -
                # or, round 6, the front part of the NEXT layer's NodeModel backward launch (DosxMlpLnBwd.pre = 1)
                if chain:
                    pending = dict(kind="node_grad", dz=dcat_e[4], rowptr_src=m.rowptr_src, perm_src=m.perm_src, aggd=aggD, w=W0,
                                   res=dcat_n[:, :H], res2=None if fold_dx else dx, aggs=aggS, flush_after=do_flush and _FLUSH_AFTER_CHAIN)
                else:
                    ops.node_grad(N, H, dcat_e[4], m.rowptr_src, m.perm_src, aggD, W0, dcat_n[:, :H], None if fold_dx else dx, aggS, dx_old)
                sink._keep.extend([dcat_n, dx, dcat_e[4], aggD])
                if src_job is not None:
                    src_job()                          # (described BEHIND the launch that writes its operand; early: next group)
                if early:
                    dx, de = dx_old, de_new
                    continue
                if l == 0 and sink.wside is not None:
                    flush_side(l)     # (layer 0's jobs read the source sums that launch has just queued)
            elif _factor_fused(m, H):
                # ONE N-row product on the two node sums, each against its own column block of W0 (DosxGemm.w_seg_off)
                if fold_dx:
                    ops.gemm(N, H, [seg(aggS), seg(aggD)], W0[:, :H], dx_old, w_layout=1, w_seg_off=H, res=dcat_n[:, :H])
                else:
                    t1 = _empty(dev, N, H)
                    ops.gemm(N, H, [seg(aggS), seg(aggD)], W0[:, :H], t1, w_layout=1, w_seg_off=H, res=dx)
                    ops.mask_residual(dcat_n[:, :H], None, t1, dx_old, None, N, H)
                    sink._keep.append(t1)
                sink._keep.extend([dcat_n, dx])
            else:
                t1, t2 = _empty(dev, N, H), _empty(dev, N, H)
                ops.gemm(N, H, [seg(aggS)], W0[:, :H], t1, w_layout=1, res=dx)
                ops.gemm(N, H, [seg(aggD)], W0[:, H:2 * H], t2, w_layout=1, res=t1)
                ops.mask_residual(dcat_n[:, :H], None, t2, dx_old, None, N, H)
                sink._keep.extend([dcat_n, t1, t2, dx])
            dx, de = dx_old, de_new
            if sink.side is not None and ((_SPLIT_LATE_FLUSH == 1 and l == 1) or (_SPLIT_LATE_FLUSH == 2 and l >= 1)):
                flush_side(l)
            continue
        ops.gather_bwd(dcat_e, dcat_n.data_ptr(), 2 * H, dx, m.rowptr_dst, m.rowptr_src, m.perm_src, None, dx_old,
                       None, N, E, H)
        sink._keep.append(dcat_n)
        dx, de = dx_old, dcat_e[:, 2 * H:]
        if sink.side is not None and ((_SPLIT_LATE_FLUSH == 1 and l == 1) or (_SPLIT_LATE_FLUSH == 2 and l >= 1)):
            # the weight gradients of the layers finished so far go to the side stream NOW, underneath the first layer's
            # backward, instead of all of them at the tail of the step where nothing else is left to overlap with
            flush_side(l)
    return dx, de


# ------------------------------------------------------------------------------------------------
# TransformerEncoder      (layers/transformer.py:46-79,120-157; layers/multihead_attention.py:49-76)
# ------------------------------------------------------------------------------------------------
def _attn_desc(Sq, Bq, Nk, Bk, H, qs, qb, x, kvhat, gam, bet) -> Attn:
    a = Attn()
    a.Sq, a.Bq, a.Nk, a.Bk, a.H = Sq, Bq, Nk, Bk, H
    a.q_stride_s, a.q_stride_b, a.flags = qs, qb, 0
    a.x, a.kvhat, a.gamma0, a.beta0 = x.data_ptr(), kvhat.data_ptr(), gam.data_ptr(), bet.data_ptr()
    return a


DROP_MASK_LOG: Optional[list] = None      # tests: set to a list to receive (prefix, layer, mask tensor) of every drawn mask
EDROP_MASK_LOG: Optional[list] = None     # ... (prefix, "x" | "k" | "v", mask) of the embed dropout (K != V path) ...
FDROP_MASK_LOG: Optional[list] = None     # ... and (prefix, layer, "res1" | "relu" | "res2", mask) of the relu / res dropout sites


def head_fused_fwd(H: int, T: int) -> bool:
    """Whether encoder_fwd(head=...) computes final LayerNorm + output layer in its last ffn_fwd launch."""
    return bool(ops.ffn_supported(H) and _FUSED_HEAD_FWD and T > 0)


def encoder_fwd(P: Params, pre: str, x: torch.Tensor, Sq: int, Bq: int, qs: int, qb: int, kvhat: torch.Tensor,
                Nk: int, Bk: int, H: int, T: int, final_ln: bool = True, drop=None, head=None, fdrop=None, key_ptr=None):
    """x: query rows (row (s,bq) at (s*qs + bq*qb)); kvhat: [Nk*Bk, H] normalised keys (stale across layers).
    drop: None or (p, seed_dev, stream_base): attention dropout in training mode (multihead_attention.py:70) - every
    layer draws its own [Bq,Sq,Nk] multiplier mask (ops.dropout_mask) that the backward re-uses.
    head = (gamma, beta, w, b, xhat [rows,H], rstd [rows], dos [Bq,Sq]) (with final_ln False, head_fused_fwd(H, T)): the
    model head - LayerNorm + H->1 output layer on the encoder's output - in the last layer's ffn_fwd epilogue; the
    encoder output itself is then not materialised (None is returned for it).
    key_ptr [Bk + 1] int32 (inference only; DosxAttn.key_ptr): crystal bk attends over its first key_ptr[bk + 1] - key_ptr[bk] key
    rows only - the batch's graph_ptr makes a batched forward equal the reference's batch-size-1 evaluation (utils.py:61-143).
    fdrop = (p_relu, p_res, seed_dev, stream_base) or None: relu / res dropout of the layer (transformer.py:137,145-147) in
    training mode.  The layer then runs UNFUSED - attention without its residual epilogue, the two feed-forward GEMMs, and
    ops.mask_residual for the three "dropout -> add residual" steps - with one explicit multiplier mask per dropout site."""
    dev = kvhat.device
    rows = Sq * Bq
    lay = []
    fin_fused = None
    if fdrop is not None and not (fdrop[0] > 0.0 or fdrop[1] > 0.0):
        fdrop = None
    assert fdrop is None or head is None
    stack = [] if _FFN_MULTI else None      # round 6: the stack's layers in one launch when every layer is row-local (see below)
    for t in range(T):
        lp = f"{pre}.layers.{t}"
        g0, b0 = P[lp + ".layer_norms.0.weight"], P[lp + ".layer_norms.0.bias"]
        x1 = _empty(dev, rows, H)
        probs = _empty(dev, Bq, Sq, Nk)
        qstats = _empty(dev, rows, 2)
        st1 = _empty(dev, rows, 2)
        a = _attn_desc(Sq, Bq, Nk, Bk, H, qs, qb, x, kvhat, g0, b0)
        a.out, a.probs, a.qstats, a.out_stats = x1.data_ptr(), probs.data_ptr(), qstats.data_ptr(), st1.data_ptr()
        if key_ptr is not None:
            assert key_ptr.dtype == torch.int32 and key_ptr.numel() >= Bk + 1 and fdrop is None and drop is None
            a.key_ptr = key_ptr.data_ptr()
        mask = None
        if drop is not None and drop[0] > 0.0:
            mask = _empty(dev, Bq, Sq, Nk)
            ops.dropout_mask(mask, drop[0], drop[1], drop[2] + t)
            a.drop_mask = mask.data_ptr()
            if DROP_MASK_LOG is not None:
                DROP_MASK_LOG.append((pre, t, mask))
        fm = None
        if fdrop is not None:
            # relu / res dropout: x1 = x + attn o M1 ; hd = relu(fc1(LN1(x1))) o M2 ; x2 = x1 + fc2(hd) o M3
            p_relu, p_res, fseed, fbase = fdrop

            def draw(shape, p, k):
                if p <= 0.0:
                    return None
                m_ = _empty(dev, *shape)
                ops.dropout_mask(m_, p, fseed, fbase + 3 * t + k)
                if FDROP_MASK_LOG is not None:
                    FDROP_MASK_LOG.append((pre, t, ("res1", "relu", "res2")[k], m_))
                return m_
            m1, m2, m3 = draw((rows, H), p_res, 0), draw((rows, 4 * H), p_relu, 1), draw((rows, H), p_res, 2)
            fm = (m1, m2, m3)
            a.flags |= 2                                   # DOSX_ATTN_NO_RESIDUAL: out = the attention output alone
            att = _empty(dev, rows, H)
            a.out = att.data_ptr()
            ops.attention_fwd(a)
            assert qs == Bq and qb == 1, "relu / res dropout: dense query rows only (layers.TransformerEncoder)"
            ops.mask_residual(att, m1, x, x1, st1, rows, H)
            h = _empty(dev, rows, 4 * H)
            ops.gemm(rows, 4 * H, [seg(x1)], P[lp + ".fc1.weight"], h, pro=PRO_ROWLN,
                     pro_gamma=P[lp + ".layer_norms.1.weight"], pro_beta=P[lp + ".layer_norms.1.bias"], pro_stats=st1,
                     bias=P[lp + ".fc1.bias"], act=ACT_RELU)
            if m2 is not None:
                ops.mask_residual(h, m2, None, h, None, rows, 4 * H)       # in place: h is the DROPPED activation from here on
            y2 = _empty(dev, rows, H)
            ops.gemm(rows, H, [seg(h)], P[lp + ".fc2.weight"], y2, bias=P[lp + ".fc2.bias"])
            x2 = _empty(dev, rows, H)
            ops.mask_residual(y2, m3, x1, x2, None, rows, H)
            lay.append((x, qs, qb, x1, probs, qstats, st1, h, mask, fm))
            x, qs, qb = x2, Bq, 1
            continue
        # <= 16 keys per crystal (the cross attention over the atoms of a crystal): the attention half runs in the prologue of
        # the feed-forward launch (DosxFfn.att_*) - one launch per layer; same saved tensors, the backward is unchanged
        # ... while the feed-forward launch runs 16-row workgroups (<= 4096 rows: one pass of the row prologue, one launch
        # 20.5 us against 6.9 + 16.4 us + a launch gap at M = 3264); with 32-row workgroups the prologue takes two passes and
        # buys nothing (33.2 vs 33.1 us at M = 6528), at 200 k rows it loses to the MFMA attention kernel (828 vs 738 us):
        # profiles/r04_kernel_microbench.log, `layer`
        att_fused = _FUSED_ATT_FFN and rows <= _ATT_FFN_MAX_ROWS and ops.ffn_att_supported(H, Nk)
        # ... and with CRYSTAL-ALIGNED tiles (round 5: a workgroup = consecutive query rows of ONE crystal, its <= 64 key rows
        # staged in LDS once, scores and P.K on the MFMA): the 51-key self attention and the 32-row launches too, while that
        # grid - Sq padded to the tile height per crystal - is one round of workgroups
        att_aligned = False
        if _FUSED_ATT_FFN and _ATT_ALIGNED and fdrop is None and ops.ffn_supported(H) and ops.ffn_att_aligned_supported(H, Nk):
            r_al = _lib_load().dosx_ffn_att_aligned_rows(int(Sq), int(Bq))      # (16 / 32: the library's own tile-height policy)
            if Bq * ((Sq + r_al - 1) // r_al) <= _ATT_ALIGNED_MAX_WGS and not (att_fused and _ATT_ROWS_FIRST):
                att_fused = att_aligned = True
        xln = None
        if not att_fused:
            if stack:                       # (this layer launches its attention on its own: what is pending goes first)
                ops.ffn_fwd_multi(stack)
                stack.clear()
            if _LN1_IN_ATTN and not ops.ffn_supported(H) and Nk <= 320:
                # unfused feed-forward half (hidden > 128): the attention kernel also writes LN1 of its output rows - they are
                # in its registers with their statistics - and fc1 reads a plain operand instead of normalising its A tile in the
                # prologue of every one of its 4H / 128 column tiles (M = 25728: 163 -> 139 us, DosxAttn.ln1_out)
                xln = _empty(dev, rows, H)
                a.ln1_gamma, a.ln1_beta = P[lp + ".layer_norms.1.weight"].data_ptr(), P[lp + ".layer_norms.1.bias"].data_ptr()
                a.ln1_out = xln.data_ptr()
                x1._dosx_ln1 = xln             # (eager mode: lives as long as the saved x1; the tail rows read it on the side stream)
            ops.attention_fwd(a)
        h = _empty(dev, rows, 4 * H)
        x2 = _empty(dev, rows, H)
        if ops.ffn_supported(H):
            # both GEMMs of the feed-forward half in one launch (the 32 x 4H intermediate tile stays in LDS); the last
            # layer also applies the encoder's final LayerNorm in its row epilogue
            fin_args = None
            if final_ln and t == T - 1:
                fin_fused = (_empty(dev, rows, H), _empty(dev, rows))
                fin_args = (P[pre + ".layer_norm.weight"], P[pre + ".layer_norm.bias"]) + fin_fused
            elif head is not None and t == T - 1:
                assert not final_ln and head_fused_fwd(H, T)
                hg, hb, hw, hbias, hxh, hrs, hdos = head
                fin_args = (hg, hb, hxh, hrs, hw, hbias, hdos, Sq, Bq)
                x2 = None
            att_args = None
            if att_fused:
                att_args = dict(kvhat=kvhat, gamma0=g0, beta0=b0, Nk=Nk, Bk=Bk, Bq=Bq, Sq=Sq, qs=qs, qb=qb, probs=probs,
                                qstats=qstats, x1=x1, st1=st1, mask=mask, aligned=att_aligned, key_ptr=key_ptr)
            ops.ffn_fwd(rows, H, x if att_fused else x1, None if att_fused else st1, P[lp + ".layer_norms.1.weight"],
                        P[lp + ".layer_norms.1.bias"], P[lp + ".fc1.weight"], P[lp + ".fc1.bias"], P[lp + ".fc2.weight"],
                        P[lp + ".fc2.bias"], h, x2, fin=fin_args, att=att_args, defer=stack if att_fused else None)
        else:
            def ffn_rows(r0, r1, x1=x1, st1=st1, h=h, x2=x2, lp=lp, xln=xln):
                if xln is not None and _bf16x3_ok(r1 - r0, 4 * H, H):
                    ops.gemm_bf16x3(xln[r0:r1], P[lp + ".fc1.weight"], h[r0:r1], bias=P[lp + ".fc1.bias"], act=ACT_RELU)
                elif xln is not None:
                    ops.gemm(r1 - r0, 4 * H, [seg(xln[r0:r1])], P[lp + ".fc1.weight"], h[r0:r1], bias=P[lp + ".fc1.bias"], act=ACT_RELU)
                else:
                    ops.gemm(r1 - r0, 4 * H, [seg(x1[r0:r1])], P[lp + ".fc1.weight"], h[r0:r1], pro=PRO_ROWLN,
                             pro_gamma=P[lp + ".layer_norms.1.weight"], pro_beta=P[lp + ".layer_norms.1.bias"], pro_stats=st1[r0:r1],
                             bias=P[lp + ".fc1.bias"], act=ACT_RELU)
                if _bf16x3_ok(r1 - r0, H, 4 * H):
                    ops.gemm_bf16x3(h[r0:r1], P[lp + ".fc2.weight"], x2[r0:r1], bias=P[lp + ".fc2.bias"], res=x1[r0:r1])
                else:
                    ops.gemm(r1 - r0, H, [seg(h[r0:r1])], P[lp + ".fc2.weight"], x2[r0:r1], bias=P[lp + ".fc2.bias"], res=x1[r0:r1])
            mt = _ffn_tail_start(rows, H)
            if mt:
                # the rows beyond the last FULL round of 64-row tiles (25728 = 3 x 8192 + 1152) as their own two-GEMM chain on
                # the side stream, next to the main rows instead of behind them as a fourth, 14 %-full round of workgroups
                ops.concurrent(dev, lambda: ffn_rows(mt, rows), lambda: ffn_rows(0, mt))
            else:
                ffn_rows(0, rows)
        lay.append((x, qs, qb, x1, probs, qstats, st1, h, mask, None))
        x, qs, qb = x2, Bq, 1
    if stack:
        # every transformer layer attends over the ORIGINAL keys (transformer.py:72-73): with the attention half inside the launch a
        # layer is row-local per tile, so the stack's layers run back to back in ONE launch (dosx_ffn_fwd_multi, two per launch)
        ops.ffn_fwd_multi(stack)
    fin = None
    if final_ln and fin_fused is not None:
        fin = fin_fused                  # x already is the normalised output
    elif final_ln:
        y = _empty(dev, rows, H)
        xhat = _empty(dev, rows, H)
        rstd = _empty(dev, rows)
        ops.layernorm(x, P[pre + ".layer_norm.weight"], P[pre + ".layer_norm.bias"], y, xhat, rstd, rows, H)
        fin = (xhat, rstd)
        x = y
    return x, (lay, fin, Sq, Bq, Nk, Bk, H, T, kvhat)


# Unfused feed-forward layers (hidden > 128), FORWARD: tail rows as a concurrent two-GEMM chain.  Electron-DOS step 7.535 ->
# 7.473 ms (four interleaved rounds, profiles/r04_ab_ffn_tail.log); the same split in the backward pass, where the weight-
# gradient groups already fill every gap, gave the gain back (7.526 ms) and is not built in.
# OPT-IN (off by default; every default program is exact fp32): the plain GEMMs of an UNFUSED feed-forward half (hidden > 128:
# fc1 forward on the LayerNorm-1 rows the attention kernel left behind, fc2 forward, fc2's input gradient with the ReLU mask) on the
# split-bf16 kernel (csrc/gemm_bf16x3.hip: three bf16 terms per fp32 element, fp32 accumulate - fp32-level accuracy, not bitwise the
# fp32-FMA chain).  bench.py reports the Electron-DOS step under it as `secondary.edos_h256_b64_split_bf16`; the oracle-live tests run
# under it with their tolerances unchanged (tests/test_gpu_models.py).  fc1's input gradient keeps dosx_gemm (LayerNorm-backward epilogue).
_FFN_BF16X3 = __import__("os").environ.get("DOSX_FFN_BF16X3", "0") == "1"


def _bf16x3_ok(rows: int, n: int, k: int) -> bool:
    return _FFN_BF16X3 and ops.gemm_bf16x3_supported(rows, n, k)


_FFN_TAIL = int(__import__("os").environ.get("DOSX_FFN_TAIL", "1"))
_FFN_TAIL_MAX = int(__import__("os").environ.get("DOSX_FFN_TAIL_MAX", "2048"))


def _ffn_tail_start(rows: int, H: int) -> int:
    """First row of the tail of an unfused feed-forward layer (0: no split): the part beyond the last full round of 256
    workgroups of 64 x 128 tiles of the H-wide GEMM (H = 256: 8192 rows), when it is small (<= DOSX_FFN_TAIL_MAX rows)."""
    gy = (H + 127) // 128
    if not _FFN_TAIL or 256 % gy:
        return 0
    per_round = 256 // gy * 64
    mt = rows // per_round * per_round
    return mt if mt > 0 and 0 < rows - mt <= _FFN_TAIL_MAX else 0


_FUSED_FFN_BWD = __import__("os").environ.get("DOSX_FUSED_FFN_BWD", "1") == "1"
_LN1_IN_ATTN = __import__("os").environ.get("DOSX_LN1_IN_ATTN", "1") == "1"
_LN1_WGRAD = __import__("os").environ.get("DOSX_LN1_WGRAD", "0") == "1"     # (measured: 7.033 vs 7.036 ms - nothing; off)
_FACTOR_EDGE_WGRAD = __import__("os").environ.get("DOSX_FACTOR_EDGE_WGRAD", "1") == "1"   # EdgeModel first Linear factored into node / edge parts
_FACTOR_FUSED = __import__("os").environ.get("DOSX_FACTOR_FUSED", "1") == "1"               # ... with the gathers / node sums inside dosx_gemm's epilogues (round 5)
_EDGE_ONE_LAUNCH = __import__("os").environ.get("DOSX_EDGE_ONE_LAUNCH", "1") == "1"       # ... and the whole EdgeModel forward as one launch (H <= 128)
_EDGE_ONE_LAUNCH_BWD = __import__("os").environ.get("DOSX_EDGE_ONE_LAUNCH_BWD", "1") == "1"   # ... and its backward
_NODE_GRAD_ONE_LAUNCH = __import__("os").environ.get("DOSX_NODE_GRAD_ONE_LAUNCH", "1") == "1"   # ... and the node side of the input gradient
_FACTOR_ONE_LAUNCH_ALWAYS = __import__("os").environ.get("DOSX_FACTOR_ONE_LAUNCH_ALWAYS", "1") == "1"   # ... at every size where they apply
# a layer's weight-gradient group is flushed BEFORE its node-side launch (dosx_node_grad: 57 workgroups) and runs under it and the
# next NodeModel backward (27 workgroups) - the two windows of the GNN backward that leave most CUs idle - instead of behind them,
# under the next EdgeModel backward (213 workgroups of a CU each); only the source block's job, which reads the source sums that
# launch makes, moves to the next group: 1.187 -> 1.173 ms per cfg2 step (three interleaved pairs, tools/exp/r5_ab_flush.sh)
_GNN_FLUSH_BEFORE_NODE = __import__("os").environ.get("DOSX_GNN_FLUSH_BEFORE_NODE", "1") == "1"
_FACTOR_MIN_GF = float(__import__("os").environ.get("DOSX_FACTOR_MIN_GF", "0.5" if _FACTOR_FUSED else "4"))
_FACTOR_DGRAD = __import__("os").environ.get("DOSX_FACTOR_DGRAD", "1") == "1"             # ... and its input gradient
_FACTOR_HEADS = __import__("os").environ.get("DOSX_FACTOR_HEADS", "1") == "1"             # heads: per-crystal K-segments multiplied once per crystal
_FACTOR_HEADS_MIN_GF = float(__import__("os").environ.get("DOSX_FACTOR_HEADS_MIN_GF", "1"))
_FACTOR_LAST = __import__("os").environ.get("DOSX_FACTOR_LAST", "1") == "1"               # last layer: aggregate, then the second Linear
_FACTOR_LAST_MIN_GF = float(__import__("os").environ.get("DOSX_FACTOR_LAST_MIN_GF", "1.3"))
_FUSED_ATT_FFN = __import__("os").environ.get("DOSX_FUSED_ATT_FFN", "1") == "1"       # <= 16-key attention inside dosx_ffn_fwd
_ATT_FFN_MAX_ROWS = int(__import__("os").environ.get("DOSX_ATT_FFN_MAX_ROWS", "4096"))
_ATT_ALIGNED = __import__("os").environ.get("DOSX_ATT_ALIGNED", "1") == "1"           # crystal-aligned tiles (DosxFfn.att_aligned)
_ATT_ALIGNED_MAX_WGS = int(__import__("os").environ.get("DOSX_ATT_ALIGNED_MAX_WGS", "256"))
_ATT_ROWS_FIRST = __import__("os").environ.get("DOSX_ATT_ROWS_FIRST", "1") == "1"     # <= 16 keys and <= 4096 rows: the per-row form
_FUSED_ATT_BWD = __import__("os").environ.get("DOSX_FUSED_ATT_BWD", "1") == "1"        # attention backward inside dosx_ffn_bwd (crystal-aligned tiles)
_FUSED_DKV = __import__("os").environ.get("DOSX_FUSED_DKV", "1") == "1"
# the self encoder's weight-gradient group starts BEHIND the small head kernels (rownorm_bwd_act, the two dE1 dgrads) instead of in
# front of them: "1" / "0", or "auto" = where the heads' weight gradients are not factored (their B-row jobs read a row sum that
# is made later, on the side stream).  Round 3: no gain; round 5, with the encoder layers' backward one launch
# each: 1.1224-1.1284 vs 1.1298-1.1331 ms per cfg2 step, three interleaved pairs (tools/exp/r5_ab_rowsfirst.sh)
_LATE_SELF_FLUSH = __import__("os").environ.get("DOSX_LATE_SELF_FLUSH", "auto")
_FUSED_FIN_BWD = __import__("os").environ.get("DOSX_FUSED_FIN_BWD", "1") == "1"
_FUSED_HEAD_FWD = __import__("os").environ.get("DOSX_FUSED_HEAD_FWD", "1") == "1"
_FUSED_HEAD_NORM = __import__("os").environ.get("DOSX_FUSED_HEAD_NORM", "1") == "1"     # DosxGemm.norm_out in the two heads


def head_fused_bwd(H: int, T: int) -> bool:
    """Whether encoder_bwd(head=...) runs the output layer + final LayerNorm backward inside its first ffn_bwd launch."""
    return bool(ops.ffn_supported(H) and _FUSED_FFN_BWD and _FUSED_FIN_BWD and T > 0)


def encoder_bwd(P: Params, G: Params, pre: str, ctx, dy: Optional[torch.Tensor], dkvhat: torch.Tensor, sink: GradSink,
                dkv_fresh: bool = False, kv_needed_next: bool = False, head=None):
    """dy: grad of the encoder output (after the final LN if there is one).  Accumulates into dkvhat
    (``dkv_fresh``: dkvhat is uninitialised and every row of it is a key row — the first layer processed
    overwrites it, which saves the zero fill).
    ``head = (gamma, beta, xhat, rstd, ddos [Bq,S], w, (G keys: ln weight, ln bias, out weight, out bias))`` with dy None
    (only when head_fused_bwd(H, T)): the encoder's output went through LN -> H->1 output layer (ops.ln_rowdot) and ddos is
    the gradient of that layer's output; their backward runs inside the last layer's ffn_bwd launch.
    Returns the gradient w.r.t. the (expanded [Sq*Bq, H]) query input."""
    lay, fin, Sq, Bq, Nk, Bk, H, T, kvhat = ctx
    dev = kvhat.device
    rows = Sq * Bq
    r32 = _rows32(rows)
    dx = dy
    fused_all = ops.ffn_supported(H) and _FUSED_FFN_BWD
    last_plain = T > 0 and lay[T - 1][9] is None            # (a layer on the relu / res dropout path runs unfused)
    fused = fused_all and last_plain
    fin_fused = None        # the final LayerNorm's backward rides in the last layer's ffn_bwd launch (csrc/ffn.hip)
    fin_keys = (pre + ".layer_norm.weight", pre + ".layer_norm.bias")
    if head is not None:
        assert dy is None and fin is None and head_fused_bwd(H, T)
        hg, hb, hx, hr, hd, hw, fin_keys = head
        dx = _empty(dev, rows, H)
        fin_fused = (hg, hx, hr, dx, hd, hw, hb, Sq, Bq)
    elif fin is not None and fused and _FUSED_FIN_BWD and T > 0 and dy.stride(1) == 1:
        dx = _empty(dev, rows, H)
        fin_fused = (P[pre + ".layer_norm.weight"], fin[0], fin[1], dx)
    elif fin is not None:
        xhat, rstd = fin
        part = sink.scratch(r32, 2 * H)
        dx = _empty(dev, rows, H)
        ops.layernorm_bwd(dy, xhat, rstd, P[pre + ".layer_norm.weight"], dx, part, rows, H)
        sink.add(part, 0, G[pre + ".layer_norm.weight"], r32, 2 * H, H)
        sink.add(part, H, G[pre + ".layer_norm.bias"], r32, 2 * H, H)
    nqt, nkt = (Sq + 31) // 32, (Nk + 31) // 32
    for t in reversed(range(T)):
        lp = f"{pre}.layers.{t}"
        x_in, qs, qb, x1, probs, qstats, st1, h, mask, fm = lay[t]
        g1, b1 = P[lp + ".layer_norms.1.weight"], P[lp + ".layer_norms.1.bias"]
        g0, b0 = P[lp + ".layer_norms.0.weight"], P[lp + ".layer_norms.0.bias"]
        dx_res2 = dx                      # gradient of x2 (the residual path of the feed-forward half carries it unchanged)
        if fm is not None and fm[2] is not None:           # x2 = x1 + y o M3: the fc2 output sees dx o M3
            dy2 = _empty(dev, rows, H)
            ops.mask_residual(dx, fm[2], None, dy2, None, rows, H)
            dx = dy2
        # fc2
        _wgrad_linear(sink, G, lp + ".fc2.weight", lp + ".fc2.bias", rows, H, seg(dx), [seg(h)], keep=(dx,))
        dh = _empty(dev, rows, 4 * H)
        dx1 = _empty(dev, rows, H)
        pld = 2 * H
        fused = fused_all and fm is None
        small = bool(_lib_load().dosx_attention_pkv_supported(int(Nk), int(H)))
        # round 5: the attention half's backward INSIDE the feed-forward half's launch (crystal-aligned tiles, DosxFfnBwd.att_*):
        # the layer's backward is one launch and dL/dx1 never reaches HBM - while that grid is one round of workgroups
        att_in_ffn = (fused and small and _FUSED_DKV and _FUSED_ATT_BWD and ops.ffn_att_bwd_supported(H, Nk, Sq, Bq)
                      and ops.ffn_att_bwd_partial_rows(Sq, Bq) <= _ATT_ALIGNED_MAX_WGS)
        att_bwd_args = None
        if att_in_ffn:
            nqt_al = ops.ffn_att_bwd_partial_rows(Sq, Bq) // Bq                # query tiles per batch entry (16- or 32-row tiles)
            npart_a = Bq * nqt_al + Bk * ((Nk + 15) // 16)
            part_a = sink.scratch(npart_a, 2 * H)
            dxin_a = _empty(dev, rows, H)
            kvp_a = sink.scratch(Bq * nqt_al * Nk, H)
            att_bwd_args = dict(x=x_in, kvhat=kvhat, gamma0=g0, beta0=b0, probs=probs, qstats=qstats, mask=mask, dxin=dxin_a,
                                partials_q=part_a.data_ptr(), partials_kv=part_a.data_ptr() + 4 * Bq * nqt_al * 2 * H, dkv_part=kvp_a,
                                dkv_cnt=ops.COUNTERS.take(dev, Bk), dkvhat=dkvhat, accumulate=0 if (dkv_fresh and t == T - 1) else 1,
                                Nk=Nk, Bk=Bk, Bq=Bq, Sq=Sq, qs=qs, qb=qb)
        if fused:           # both dgrad GEMMs + ReLU mask + LN1 backward + residual in one launch (csrc/ffn.hip)
            rgp = ops.ffn_att_bwd_partial_rows(Sq, Bq) if att_in_ffn else ops.ffn_bwd_partial_rows(rows)
            with_fin = fin_fused is not None and t == T - 1
            pld = (5 * H + 4 if head is not None else 4 * H) if with_fin else 2 * H
            part = sink.scratch(rgp, pld)
            ops.ffn_bwd(rows, H, dy if with_fin else dx, h, x1, st1, g1, P[lp + ".fc1.weight"], P[lp + ".fc2.weight"], dh,
                        None if att_in_ffn else dx1, part, fin=fin_fused if with_fin else None, att=att_bwd_args)
            if with_fin:
                sink.add(part, 2 * H, G[fin_keys[0]], rgp, pld, H)
                sink.add(part, 3 * H, G[fin_keys[1]], rgp, pld, H)
                if head is not None:
                    sink.add(part, 4 * H, G[fin_keys[2]], rgp, pld, H)
                    sink.add(part, 5 * H, G[fin_keys[3]], rgp, pld, 1)
        else:
            if _bf16x3_ok(rows, 4 * H, H) and dx.is_contiguous():
                ops.gemm_bf16x3(dx, P[lp + ".fc2.weight"], dh, w_layout=1, mask=h)
            else:
                ops.gemm(rows, 4 * H, [seg(dx)], P[lp + ".fc2.weight"], dh, w_layout=1, epi=EPI_RELU_MASK, aux=h)
            if fm is not None and fm[1] is not None:       # h is the dropped activation: [h > 0] = [relu > 0][M2 > 0]; x 1/(1-p)
                ops.mask_residual(dh, fm[1], None, dh, None, rows, 4 * H)
        # fc1 (+ LN1 backward + residual)
        xln = getattr(x1, "_dosx_ln1", None) if _LN1_WGRAD else None
        if xln is not None:         # the forward left LN1(x1) behind (DosxAttn.ln1_out): a plain operand for the weight gradient too
            _wgrad_linear(sink, G, lp + ".fc1.weight", lp + ".fc1.bias", rows, 4 * H, seg(dh), [seg(xln)], keep=(dh, xln))
        else:
            _wgrad_linear(sink, G, lp + ".fc1.weight", lp + ".fc1.bias", rows, 4 * H, seg(dh), [seg(x1)], keep=(dh,),
                          pro=PRO_ROWLN, pro_gamma=g1, pro_beta=b1, pro_stats=st1)
        if not fused:
            rgp = ops.gemm_partial_rows(rows, H, EPI_ROWLN_BWD)
            part = sink.scratch(rgp, 2 * H)
            ops.gemm(rows, H, [seg(dh)], P[lp + ".fc1.weight"], dx1, w_layout=1, epi=EPI_ROWLN_BWD, aux=x1, aux_stats=st1,
                     epi_gamma=g1, res=dx_res2, partials=part, partial_ld=2 * H)
        sink.add(part, 0, G[lp + ".layer_norms.1.weight"], rgp, pld, H)
        sink.add(part, H, G[lp + ".layer_norms.1.bias"], rgp, pld, H)
        # attention (+ LN0 backward on the query side + residual); key side accumulates into dkvhat
        # Nk <= 64 (atoms of a crystal, the 51 phonon bins): the dq kernel leaves every query tile's share of dK + dV in
        # `kvp` and the dk+dv half is a small reduction over those partials; larger key sets (201 eDOS bins) stream the
        # dS round trip through `dsc` into the dkv kernel
        if att_in_ffn:          # (the attention half ran inside the ffn_bwd launch above)
            sink.add(part_a, 0, G[lp + ".layer_norms.0.weight"], npart_a, 2 * H, H)
            sink.add(part_a, H, G[lp + ".layer_norms.0.bias"], npart_a, 2 * H, H)
            sink._keep.extend(t_ for t_ in (mask, x_in) if t_ is not None)
            dx = dxin_a
            continue
        npart = Bq * nqt + Bk * ((Nk + 15) // 16 if small else nkt)        # key-side partial rows: per 16 / 32 keys
        part = sink.scratch(npart, 2 * H)
        dxin = _empty(dev, rows, H)
        dsc = None if small else _empty(dev, Bq, Sq, Nk)
        kvp = sink.scratch(Bq * nqt * Nk, H) if small else None
        acc = 0 if (dkv_fresh and t == T - 1) else 1

        dout_att = dx1
        if fm is not None:                 # x1 = x + attn o M1: the attention output sees dx1 o M1, x the plain dx1 (added below)
            dout_att = _empty(dev, rows, H)
            ops.mask_residual(dx1, fm[0], None, dout_att, None, rows, H)

        def desc(flags):
            a = _attn_desc(Sq, Bq, Nk, Bk, H, qs, qb, x_in, kvhat, g0, b0)
            a.probs, a.qstats = probs.data_ptr(), qstats.data_ptr()
            a.dout, a.dx, a.dkvhat, a.dkv_accumulate = dout_att.data_ptr(), dxin.data_ptr(), dkvhat.data_ptr(), acc
            if fm is not None:
                flags |= 2                 # DOSX_ATTN_NO_RESIDUAL
            a.dscores = dsc.data_ptr() if dsc is not None else None
            a.dkv_part = kvp.data_ptr() if kvp is not None else None
            a.drop_mask = mask.data_ptr() if mask is not None else None
            a.partials_q = part.data_ptr()
            a.partials_kv = part.data_ptr() + 4 * Bq * nqt * 2 * H
            a.flags = flags
            return a
        if small and _FUSED_DKV:
            # ONE launch: every (query tile, crystal) workgroup publishes its share of dK + dV and the last one of a key
            # crystal to arrive finishes that crystal's key gradient (include/dosx.h: DosxAttn.dkv_cnt) - no reduction
            # launch, no side-stream round trip
            a1 = desc(0)
            a1.dkv_cnt = ops.COUNTERS.take(dev, Bk)
            ops.attention_bwd(a1)
            sink._keep.extend(t_ for t_ in (mask,) if t_ is not None)
        else:
            # dq (feeds the next layer's backward) on the main stream; dk+dv (feeds only the key-gradient
            # consumers at the very end) on the side stream, in layer order so dkvhat accumulates in order
            ops.attention_bwd(desc(8))          # DOSX_ATTN_BWD_DQ_HALF
            a2 = desc(4)                        # DOSX_ATTN_BWD_DKV_HALF
            if kv_needed_next and t == 0:
                # the caller consumes dkvhat right after this call: the LAST key-gradient kernel runs on the main stream
                # (behind a join that is already satisfied - the earlier layers' reductions finished long ago) instead of
                # bouncing main -> side -> main through two cross-queue events
                sink.join()
                ops.attention_bwd(a2)
                sink._keep.extend(t_ for t_ in (dsc, mask) if t_ is not None)
            else:
                sink.on_side(lambda a2=a2: ops.attention_bwd(a2), (dx1, dxin) + tuple(t_ for t_ in (dsc, mask) if t_ is not None))
        sink.add(part, 0, G[lp + ".layer_norms.0.weight"], npart, 2 * H, H)
        sink.add(part, H, G[lp + ".layer_norms.0.bias"], npart, 2 * H, H)
        if fm is not None:                 # + the residual path of the attention half
            sink._keep.append(dout_att)
            dsum = _empty(dev, rows, H)
            ops.mask_residual(dxin, None, dx1, dsum, None, rows, H)
            dxin = dsum
        dx = dxin
    return dx


# ------------------------------------------------------------------------------------------------
# TransformerEncoder with K != V  (x_in_k is not x_in_v, or embed dropout: transformer.py:61-68 draws different masks for keys
# and values).  No reference call site does this, so it is a correctness path: every layer runs unfused - three LayerNorm
# launches (layer_norms[0] is shared by q, k and v, transformer.py:131-134), the softmax weights from the MFMA attention
# kernel run on the keys, P.V and the whole attention backward from the building blocks of csrc/attention_kv.hip, the
# feed-forward half as two GEMMs - with the four dropout sites as explicit multiplier masks.
# ------------------------------------------------------------------------------------------------
def encoder_kv_fwd(P: Params, pre: str, x: torch.Tensor, xk: torch.Tensor, xv: torch.Tensor, Sq: int, Bq: int, Nk: int, Bk: int,
                   H: int, T: int, final_ln: bool = True, drop=None, fdrop=None):
    """x [Sq*Bq, H] dense query rows, xk / xv [Nk*Bk, H] (the ORIGINAL keys / values: not updated between layers,
    transformer.py:72-73).  drop = (p, seed, base) attention dropout, fdrop = (p_relu, p_res, seed, base)."""
    dev = x.device
    rows, krows = Sq * Bq, Nk * Bk
    ones, zeros = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    lay = []
    for t in range(T):
        lp = f"{pre}.layers.{t}"
        g0, b0 = P[lp + ".layer_norms.0.weight"], P[lp + ".layer_norms.0.bias"]

        def ln(src, n):
            y, xh, rs = _empty(dev, n, H), _empty(dev, n, H), _empty(dev, n)
            ops.layernorm(src, g0, b0, y, xh, rs, n, H)
            return y, xh, rs
        q, qh, qr = ln(x, rows)
        k, kh, kr = ln(xk, krows)
        v, vh, vr = ln(xv, krows)
        probs = _empty(dev, Bq, Sq, Nk)
        if H <= ops.ATTN_MAX_H:
            scratch = _empty(dev, rows, H)
            a = _attn_desc(Sq, Bq, Nk, Bk, H, Bq, 1, q, k, ones, zeros)
            a.flags = 1 | 2                                # RAW_Q | NO_RESIDUAL: q and k are already normalised
            a.out, a.probs = scratch.data_ptr(), probs.data_ptr()
            ops.attention_fwd(a)                           # (only the softmax weights are used)
        else:                                              # wider rows: scores + row softmax
            scores = _empty(dev, Bq, Sq, Nk)
            ops.attn_dp(q, k, scores, Sq, Bq, Nk, Bk, H)
            ops.softmax_fwd(scores, probs, Bq * Sq, Nk, H ** -0.5)
        amask = None
        if drop is not None and drop[0] > 0.0:
            amask = _empty(dev, Bq, Sq, Nk)
            ops.dropout_mask(amask, drop[0], drop[1], drop[2] + t)
            if DROP_MASK_LOG is not None:
                DROP_MASK_LOG.append((pre, t, amask))
        att = _empty(dev, rows, H)
        ops.attn_pv(probs, amask, v, att, Sq, Bq, Nk, Bk, H)
        m1 = m2 = m3 = None
        if fdrop is not None:
            p_relu, p_res, fseed, fbase = fdrop

            def draw(shape, p, kk):
                if p <= 0.0:
                    return None
                m_ = _empty(dev, *shape)
                ops.dropout_mask(m_, p, fseed, fbase + 3 * t + kk)
                if FDROP_MASK_LOG is not None:
                    FDROP_MASK_LOG.append((pre, t, ("res1", "relu", "res2")[kk], m_))
                return m_
            m1, m2, m3 = draw((rows, H), p_res, 0), draw((rows, 4 * H), p_relu, 1), draw((rows, H), p_res, 2)
        x1, st1 = _empty(dev, rows, H), _empty(dev, rows, 2)
        ops.mask_residual(att, m1, x, x1, st1, rows, H)
        h = _empty(dev, rows, 4 * H)
        ops.gemm(rows, 4 * H, [seg(x1)], P[lp + ".fc1.weight"], h, pro=PRO_ROWLN, pro_gamma=P[lp + ".layer_norms.1.weight"],
                 pro_beta=P[lp + ".layer_norms.1.bias"], pro_stats=st1, bias=P[lp + ".fc1.bias"], act=ACT_RELU)
        if m2 is not None:
            ops.mask_residual(h, m2, None, h, None, rows, 4 * H)
        y2, x2 = _empty(dev, rows, H), _empty(dev, rows, H)
        ops.gemm(rows, H, [seg(h)], P[lp + ".fc2.weight"], y2, bias=P[lp + ".fc2.bias"])
        ops.mask_residual(y2, m3, x1, x2, None, rows, H)
        lay.append((q, qh, qr, k, kh, kr, v, vh, vr, probs, amask, x1, st1, h, (m1, m2, m3)))
        x = x2
    fin = None
    if final_ln:
        y, xhat, rstd = _empty(dev, rows, H), _empty(dev, rows, H), _empty(dev, rows)
        ops.layernorm(x, P[pre + ".layer_norm.weight"], P[pre + ".layer_norm.bias"], y, xhat, rstd, rows, H)
        fin, x = (xhat, rstd), y
    return x, (lay, fin, Sq, Bq, Nk, Bk, H, T)


def encoder_kv_bwd(P: Params, G: Params, pre: str, ctx, dy: torch.Tensor, sink: GradSink):
    """Returns (dx [Sq*Bq,H], dxk [Nk*Bk,H], dxv [Nk*Bk,H])."""
    lay, fin, Sq, Bq, Nk, Bk, H, T = ctx
    dev = dy.device
    rows, krows = Sq * Bq, Nk * Bk
    r32, k32 = _rows32(rows), _rows32(krows)
    dx = dy
    if fin is not None:
        part = sink.scratch(r32, 2 * H)
        dx = _empty(dev, rows, H)
        ops.layernorm_bwd(dy, fin[0], fin[1], P[pre + ".layer_norm.weight"], dx, part, rows, H)
        sink.add(part, 0, G[pre + ".layer_norm.weight"], r32, 2 * H, H)
        sink.add(part, H, G[pre + ".layer_norm.bias"], r32, 2 * H, H)
    dxk = dxv = None
    for t in reversed(range(T)):
        lp = f"{pre}.layers.{t}"
        q, qh, qr, k, kh, kr, v, vh, vr, probs, amask, x1, st1, h, (m1, m2, m3) = lay[t]
        g0 = P[lp + ".layer_norms.0.weight"]
        g1, b1 = P[lp + ".layer_norms.1.weight"], P[lp + ".layer_norms.1.bias"]
        dx2 = dx
        dyf = dx2
        if m3 is not None:
            dyf = _empty(dev, rows, H)
            ops.mask_residual(dx2, m3, None, dyf, None, rows, H)
        _wgrad_linear(sink, G, lp + ".fc2.weight", lp + ".fc2.bias", rows, H, seg(dyf), [seg(h)], keep=(dyf,))
        dh = _empty(dev, rows, 4 * H)
        ops.gemm(rows, 4 * H, [seg(dyf)], P[lp + ".fc2.weight"], dh, w_layout=1, epi=EPI_RELU_MASK, aux=h)
        if m2 is not None:
            ops.mask_residual(dh, m2, None, dh, None, rows, 4 * H)
        _wgrad_linear(sink, G, lp + ".fc1.weight", lp + ".fc1.bias", rows, 4 * H, seg(dh), [seg(x1)], keep=(dh,),
                      pro=PRO_ROWLN, pro_gamma=g1, pro_beta=b1, pro_stats=st1)
        rgp = ops.gemm_partial_rows(rows, H, EPI_ROWLN_BWD)
        part = sink.scratch(rgp, 2 * H)
        dx1 = _empty(dev, rows, H)
        ops.gemm(rows, H, [seg(dh)], P[lp + ".fc1.weight"], dx1, w_layout=1, epi=EPI_ROWLN_BWD, aux=x1, aux_stats=st1,
                 epi_gamma=g1, res=dx2, partials=part, partial_ld=2 * H)
        sink.add(part, 0, G[lp + ".layer_norms.1.weight"], rgp, 2 * H, H)
        sink.add(part, H, G[lp + ".layer_norms.1.bias"], rgp, 2 * H, H)
        datt = dx1
        if m1 is not None:
            datt = _empty(dev, rows, H)
            ops.mask_residual(dx1, m1, None, datt, None, rows, H)
        # attention: dV = (P o M)^T dAtt ; dP = dAtt V^T ; dS = softmax'(P, dP o M) ; dQ = dS K ; dK = dS^T Q
        dv, dk, dq = _empty(dev, krows, H), _empty(dev, krows, H), _empty(dev, rows, H)
        dpd, ds = _empty(dev, Bq, Sq, Nk), _empty(dev, Bq, Sq, Nk)
        ops.attn_tv(probs, amask, datt, dv, Sq, Bq, Nk, Bk, H)
        ops.attn_dp(datt, v, dpd, Sq, Bq, Nk, Bk, H)
        ops.softmax_bwd(probs, amask, dpd, ds, Bq * Sq, Nk, H ** -0.5)
        ops.attn_pv(ds, None, k, dq, Sq, Bq, Nk, Bk, H)
        ops.attn_tv(ds, None, q, dk, Sq, Bq, Nk, Bk, H)
        # the shared LayerNorm 0 behind q, k and v: three backward launches, their parameter-gradient partials accumulate
        pq, pk, pv = sink.scratch(r32, 2 * H), sink.scratch(k32, 2 * H), sink.scratch(k32, 2 * H)
        dxq, dk_in, dv_in = _empty(dev, rows, H), _empty(dev, krows, H), _empty(dev, krows, H)
        ops.layernorm_bwd(dq, qh, qr, g0, dxq, pq, rows, H)
        ops.layernorm_bwd(dk, kh, kr, g0, dk_in, pk, krows, H)
        ops.layernorm_bwd(dv, vh, vr, g0, dv_in, pv, krows, H)
        for prt, n32 in ((pq, r32), (pk, k32), (pv, k32)):
            sink.add(prt, 0, G[lp + ".layer_norms.0.weight"], n32, 2 * H, H)
            sink.add(prt, H, G[lp + ".layer_norms.0.bias"], n32, 2 * H, H)
        dsum = _empty(dev, rows, H)
        ops.mask_residual(dxq, None, dx1, dsum, None, rows, H)          # + the residual path of the attention half
        dx = dsum
        if dxk is None:
            dxk, dxv = dk_in, dv_in
        else:                                                           # the same keys / values feed every layer
            nk_, nv_ = _empty(dev, krows, H), _empty(dev, krows, H)
            ops.mask_residual(dk_in, None, dxk, nk_, None, krows, H)
            ops.mask_residual(dv_in, None, dxv, nv_, None, krows, H)
            dxk, dxv = nk_, nv_
        sink._keep.extend([datt, dv, dk, dq, dpd, ds, dxq, dk_in, dv_in, dx1, dyf])
    return dx, dxk, dxv


# ------------------------------------------------------------------------------------------------
# Full models
# ------------------------------------------------------------------------------------------------
class ModelCfg:
    def __init__(self, kind: str, n_layers: int, n_t: int, hidden: int, n_atom: int, n_bond: int, bins: int,
                 mean: bool, prompt_key: str):
        self.kind, self.L, self.T, self.H = kind, n_layers, n_t, hidden
        self.Fa, self.Fb, self.S, self.mean, self.prompt_key = n_atom, n_bond, bins, mean, prompt_key


def _f32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    if ops.RECORDER.active:
        # a torch cast is not a libdosx call: it would run once while recording and never again on replay, so the
        # replayed kernels would keep reading the first batch's converted copy (train._Slot stores fp32 for this reason)
        raise RuntimeError(f"recorded programs need float32 contiguous inputs, got {t.dtype} (contiguous={t.is_contiguous()})")
    return t.to(torch.float32).contiguous()


def _i32(t: torch.Tensor) -> torch.Tensor:
    if t.dtype == torch.int32 and t.is_contiguous():
        return t
    if ops.RECORDER.active:
        raise RuntimeError(f"recorded programs need int32 contiguous indices, got {t.dtype}")
    return t.to(torch.int32).contiguous()


def _edge_inputs(cfg: ModelCfg, g, m: GraphMeta):
    """Edge features in the kernels' (destination-sorted) edge order."""
    if cfg.kind == "phonon":
        vec = _f32(g.edge_vec)
        if m.edge_perm is not None:
            vec = vec[m.edge_perm]
        return ops.edge_feat_sh1(vec, 4.0)            # r_max = 4 (DOSTransformer_phonon.py:77)
    ea = _f32(g.edge_attr)
    if m.edge_perm is not None:
        ea = ea[m.edge_perm]
    return ea


def gnn_trunk_fwd(P: Params, cfg: ModelCfg, g, m: GraphMeta, node_key: str = "GN_encoder.node_encoder"):
    """Encoder + L processors (+ eDOS global encoder).  Returns x_L [N,H], u [B,H] or None, ctx."""
    with ops.graph_rows():
        return _gnn_trunk_fwd(P, cfg, g, m, node_key)


def _gnn_trunk_fwd(P: Params, cfg: ModelCfg, g, m: GraphMeta, node_key: str):
    H, N, E, B = cfg.H, m.num_nodes, m.num_edges, m.num_graphs
    xin = _f32(g.x)
    pq0 = None
    if (_ENC_CS and cfg.L >= 1 and m.seg_tile is not None and _factor_edge(E, H, m) and xin.shape[1] == P[node_key + ".0.weight"].shape[1]
            and ops.enc_cs_supported(N, xin.shape[1], H)):
        # round 6: Linear -> PReLU -> Linear of the node encoder AND layer 0's node products in one column-split launch
        z0, x0, pq0 = _empty(dev_of(P), N, H), _empty(dev_of(P), N, H), _empty(dev_of(P), N, 4 * H)
        ops.enc_cs_fwd(N, xin, P[node_key + ".0.weight"], P[node_key + ".0.bias"], P[node_key + ".1.weight"], P[node_key + ".2.weight"],
                       P[node_key + ".2.bias"], z0, x0, P["stacked_processor.0.edge_model.edge_mlp.0.weight"], pq0)
        cn = (SegList([seg(xin)], [xin]), z0, N, H)
    else:
        x0, cn = mlp_prelu_fwd(P, node_key, SegList([seg(xin)], [xin]), N, H)
    if cfg.kind == "phonon" and P["GN_encoder.edge_encoder.0.weight"].shape[1] == 4:
        # SH(l<=1) * cutoff features (r_max = 4, DOSTransformer_phonon.py:77) and the K = 4 Linear on them in one launch
        vec = _f32(g.edge_vec)
        if m.edge_perm is not None:
            vec = vec[m.edge_perm]
        ek = "GN_encoder.edge_encoder"
        if _EDGE_ENC_ONE_LAUNCH and ops.edge_enc_supported(H):
            # round 6: features, both Linear layers and the PReLU between them in ONE launch (csrc/heads.hip: edge_enc_fwd_kernel)
            ea, z0, e0 = ops.edge_enc_fwd(vec, P[ek + ".0.weight"], P[ek + ".0.bias"], P[ek + ".1.weight"], P[ek + ".2.weight"], P[ek + ".2.bias"], 4.0)
            ce = (SegList([seg(ea)], [ea]), z0, E, H)
        else:
            ea, z0 = ops.edge_embed_sh1(vec, P[ek + ".0.weight"], P[ek + ".0.bias"], 4.0)
            e0, ce = mlp_prelu_fwd(P, ek, SegList([seg(ea)], [ea]), E, H, z=z0)
    else:
        ea = _edge_inputs(cfg, g, m)
        e0, ce = mlp_prelu_fwd(P, "GN_encoder.edge_encoder", SegList([seg(ea)], [ea]), E, H)
    u, cu = None, None
    if cfg.kind == "edos":
        glob = _f32(g.glob).reshape(B, 2)
        u, cu = mlp_prelu_fwd(P, "GN_encoder.global_encoder", SegList([seg(glob)], [glob]), B, H)

    xL, cg = gnn_fwd(P, m, x0, e0, cfg.L, cfg.mean, H, pq0=pq0)
    return xL, u, (cn, ce, cu, cg, node_key)


def gnn_trunk_bwd(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, ctx, dxL: torch.Tensor, du_seg: Optional[Seg],
                  sink: GradSink, dx_pre=None):
    with ops.graph_rows():
        _gnn_trunk_bwd(P, G, cfg, m, ctx, dxL, du_seg, sink, dx_pre)


def _gnn_trunk_bwd(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, ctx, dxL: torch.Tensor, du_seg: Optional[Seg],
                   sink: GradSink, dx_pre=None):
    cn, ce, cu, cg, node_key = ctx
    dx0, de0 = gnn_bwd(P, G, m, cg, dxL, sink, cfg.L, cfg.mean, cfg.H, dx_pre=dx_pre)
    if _ENC_BWD_PAIR and de0 is not None and cn[3] == ce[3]:
        mlp_prelu_bwd_pair(P, G, (node_key, cn, dx0), ("GN_encoder.edge_encoder", ce, de0), sink, tail=True)
        if cu is not None and du_seg is not None:
            mlp_prelu_bwd(P, G, "GN_encoder.global_encoder", cu, None, sink, dy_seg=du_seg, tail=True)
        return
    mlp_prelu_bwd(P, G, node_key, cn, dx0, sink, tail=True)
    # (measured: the edge encoder's backward moved ahead of the first layer's gather backward, where dL/de_0 already exists,
    #  so that its weight gradients join layer 0's group instead of the tail: on the side stream +15 us (round 2), on the main
    #  stream +5 us at cfg2 and +90 us for eDOS (round 3, tools/exp/ab_early.sh) - at the tail they run as many short
    #  workgroups on an idle GPU, in the group they queue behind layer 0's long ones)
    mlp_prelu_bwd(P, G, "GN_encoder.edge_encoder", ce, de0, sink, tail=True)
    if cu is not None and du_seg is not None:
        mlp_prelu_bwd(P, G, "GN_encoder.global_encoder", cu, None, sink, dy_seg=du_seg, tail=True)


def decoder_fwd(P: Params, cfg: ModelCfg, m: GraphMeta, xL: torch.Tensor, u: Optional[torch.Tensor]):
    """GN_decoder: Linear on sum-pooled nodes (phonon) / on cat[u, pooled] (eDOS)."""
    H, B = cfg.H, m.num_graphs
    dev = xL.device
    pooled = _empty(dev, B, H)
    ops.graph_pool(xL, m.graph_ptr, pooled.data_ptr(), H, B, H)
    segs = SegList([seg(pooled)], [pooled]) if u is None else SegList([seg(u), seg(pooled)], [u, pooled])
    graph = _empty(dev, B, H)
    ops.gemm(B, H, segs.segs, P["GN_decoder.mlp.0.weight"], graph, bias=P["GN_decoder.mlp.0.bias"])
    return graph, segs


def decoder_dgrad(P: Params, cfg: ModelCfg, m: GraphMeta, segs: SegList, dgraph: torch.Tensor) -> torch.Tensor:
    """dL/d[u | pooled] of the decoder's Linear, [B, K] (the pooled block is its last H columns)."""
    dcat = _empty(dgraph.device, m.num_graphs, segs.K)
    ops.gemm(m.num_graphs, segs.K, [seg(dgraph)], P["GN_decoder.mlp.0.weight"], dcat, w_layout=1)
    return dcat


def decoder_bwd(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, segs: SegList, dgraph: torch.Tensor, dxL: Optional[torch.Tensor],
                sink: GradSink, dcat: Optional[torch.Tensor] = None) -> Optional[Seg]:
    """Adds the pooled-node gradient into dxL (dxL None: the caller folds it into another launch, see
    ops.dense_normalize_pool_bwd; dcat: the dgrad already computed with decoder_dgrad); returns the Seg of du (eDOS) or None."""
    H, B, N = cfg.H, m.num_graphs, m.num_nodes
    _wgrad_linear(sink, G, "GN_decoder.mlp.0.weight", "GN_decoder.mlp.0.bias", B, H, seg(dgraph), segs.segs, keep=(dgraph,))
    K = segs.K
    if dcat is None:
        dcat = decoder_dgrad(P, cfg, m, segs, dgraph)
    if dxL is not None:
        ops.graph_pool_bwd(dcat.data_ptr() + 4 * (K - H), K, m.node_graph, dxL, N, H, True, num_graphs=B)   # ghost nodes: zero
    sink._keep.append(dcat)
    return seg(dcat, width=H, col=0) if K == 2 * H else None


def dostransformer_fwd(P: Params, cfg: ModelCfg, g, m: GraphMeta, drop=None, per_crystal_keys: bool = False):
    """Forward of DOSTransformer_phonon / DOSTransformer (DOSTransformer_phonon.py:66-119,
    DOSTransformer.py:45-93).  Returns (dos [2B,S] : rows [0,B) global, [B,2B) system; x_L; ctx).
    drop: None (eval mode / attn_drop 0) or (p, seed_dev): attention dropout of the three encoders.
    per_crystal_keys (inference only): the two cross attentions attend over each crystal's OWN atoms instead of the batch's
    zero-padded Nmax rows - what the reference computes at batch size 1, its evaluation setting (main_eDOS.py:55-56,
    utils.py:61-143; SURVEY.md 0.3: the padded rows take part in the softmax, so outputs depend on the batch's Nmax)."""
    dr = (lambda base: None) if drop is None else (lambda base: (drop[0], drop[1], base))
    H, S, T, B, N = cfg.H, cfg.S, cfg.T, m.num_graphs, m.num_nodes
    if per_crystal_keys and (drop is not None or H > ops.ATTN_MAX_H or m.n_max > 320):
        from ._lib import DosxError
        raise DosxError("per_crystal_keys: inference mode (no dropout), hidden <= 256 and at most 320 atoms per crystal")
    kp = m.graph_ptr if per_crystal_keys else None
    if H > ops.ATTN_MAX_H:
        return _dostransformer_fwd_wide(P, cfg, g, m, dr)
    nmax = m.n_max
    dev = P["embeddings.weight"].device
    xL, u, ctrunk = gnn_trunk_fwd(P, cfg, g, m)
    # to_dense_batch + the (parameter-free part of the) key LayerNorm, shared by every cross attention
    kvhat = _empty(dev, nmax * B + 1, H)          # + 1 spare row: the dense slot of ghost (padding) nodes
    rstd_n = _empty(dev, N)
    ops.dense_normalize_slots(xL, m.graph_ptr, kvhat, rstd_n, B, nmax, H)
    emb = P["embeddings.weight"]
    # The pooled decoder input and the prompt rows do not depend on the first encoder: in a recorded program they run
    # on the side stream underneath it (three launch-latency-bound kernels off the critical path); eagerly they run here.
    sysidx = _i32(g.system)
    hp = H // 2
    side = ops.GradSink(dev)
    box = {}

    fh = _factor_heads(S * B, H)

    def _decoder_branch():
        box["graph"], box["segs"] = decoder_fwd(P, cfg, m, xL, u)
        box["prow"] = _empty(dev, B, hp)
        ops.embed_rows(P[cfg.prompt_key], sysidx, box["prow"], B, hp)
        if fh:
            # the heads read cat[E1, graph(, prompt)] with the crystal's pooled vector (and prompt row) repeated for every energy:
            # those K-segments are multiplied ONCE per crystal here (B rows) and enter the per-energy GEMM as a pre-activation
            # row term (DosxGemm.res_pre) - the heads' K shrinks from 2H / 2.5H to H
            box["qg"], box["qs"] = _empty(dev, B, H), _empty(dev, B, H)
            ops.gemm(B, H, [seg(box["graph"])], P["fc.weight"][:, H:], box["qg"])
            ops.gemm(B, H, [seg(box["graph"]), seg(box["prow"])], P["fc_prompt.weight"][:, H:], box["qs"])
    side.on_side(_decoder_branch)
    E1, c1 = encoder_fwd(P, "transformer", emb, S, B, 1, 0, kvhat, nmax, B, H, T, drop=dr(0), key_ptr=kp)
    side.join()
    graph, dec_segs, prow = box["graph"], box["segs"], box["prow"]
    dosin = _empty(dev, S * 2 * B, H)
    modB = rowmap(d=B, m=0, c=1)
    a_g = SegList([seg(E1), seg(graph, rmap=modB)], [E1, graph])
    a_s = SegList([seg(E1), seg(graph, rmap=modB), seg(prow, rmap=modB)], [E1, graph, prow])
    # (the two heads write disjoint row sets of dosin; running fc_prompt on the side stream next to fc was measured: the two
    #  cross-queue events cost more than the 11 us GEMM they hide - 1.385 vs 1.370 ms per step)
    # the self-attention encoder's stale keys are the NORMALISED head outputs: the heads' epilogues write them next to the
    # plain rows (DosxGemm.norm_out) - no dosx_rownorm launch between the heads and the encoder
    kvs = _empty(dev, S * 2 * B, H)
    rstd_s = _empty(dev, S * 2 * B)
    nk = dict(norm_out=kvs, norm_rstd=rstd_s) if _FUSED_HEAD_NORM else {}
    if fh:
        hg = dict(segs=[seg(E1)], w=P["fc.weight"][:, :H], res=box["qg"], res_map=modB, res_pre=True)
        hs = dict(segs=[seg(E1)], w=P["fc_prompt.weight"][:, :H], res=box["qs"], res_map=modB, res_pre=True)
    else:
        hg, hs = dict(segs=a_g.segs, w=P["fc.weight"]), dict(segs=a_s.segs, w=P["fc_prompt.weight"])
    ops.gemm_pair(dict(M=S * B, N=H, out=dosin, bias=P["fc.bias"], act=ACT_LEAKY, act_slope=0.01,
                       out_map=rowmap(d=B, m=2 * B, c=1, off=0), **hg, **nk),
                  dict(M=S * B, N=H, out=dosin, bias=P["fc_prompt.bias"], act=ACT_LEAKY,
                       act_slope=0.01, out_map=rowmap(d=B, m=2 * B, c=1, off=B), **hs, **nk))
    if not _FUSED_HEAD_NORM:
        ops.rownorm(dosin, kvs, rstd_s, S * 2 * B, H)
    hs, c2 = encoder_fwd(P, "transformer_self", dosin, S, 2 * B, 2 * B, 1, kvs, S, 2 * B, H, T, drop=dr(64))
    xhat_f = _empty(dev, S * 2 * B, H)
    rstd_f = _empty(dev, S * 2 * B)
    dos = _empty(dev, 2 * B, S)
    gf, bf = P["transformer_source.layer_norm.weight"], P["transformer_source.layer_norm.bias"]
    if head_fused_fwd(H, T):     # final LayerNorm + out_layer in the last ffn_fwd launch of the source encoder
        _, c3 = encoder_fwd(P, "transformer_source", hs, S, 2 * B, 2 * B, 1, kvhat, nmax, B, H, T, final_ln=False,
                            drop=dr(128), head=(gf, bf, P["out_layer.weight"], P["out_layer.bias"], xhat_f, rstd_f, dos), key_ptr=kp)
    else:
        hsrc, c3 = encoder_fwd(P, "transformer_source", hs, S, 2 * B, 2 * B, 1, kvhat, nmax, B, H, T, final_ln=False,
                               drop=dr(128), key_ptr=kp)
        ops.ln_rowdot(hsrc, gf, bf, P["out_layer.weight"], P["out_layer.bias"], xhat_f, rstd_f, dos, S, 2 * B, H)
    a_g.keep.extend(t for t in (box.get("qg"), box.get("qs")) if t is not None)
    ctx = (ctrunk, kvhat, rstd_n, c1, dec_segs, sysidx, prow, dosin, a_g, a_s, kvs, rstd_s, c2, c3, xhat_f, rstd_f, xL)
    return dos, xL, ctx


def dostransformer_bwd(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, ctx, ddos: torch.Tensor,
                       dx_ext: Optional[torch.Tensor], sink: GradSink, mid_hook=None) -> None:
    """Backward; ddos [2B,S] (rows [0,B): d dos_global, [B,2B): d dos_system); dx_ext: optional
    gradient w.r.t. the returned node embeddings.  Writes every live parameter gradient into G."""
    if ctx[0] == "wide":
        return _dostransformer_bwd_wide(P, G, cfg, m, ctx, ddos, dx_ext, sink, mid_hook)
    (ctrunk, kvhat, rstd_n, c1, dec_segs, sysidx, prow, dosin, a_g, a_s, kvs, rstd_s, c2, c3, xhat_f, rstd_f, xL) = ctx
    H, S, B, N = cfg.H, cfg.S, m.num_graphs, m.num_nodes
    nmax = m.n_max
    dev = ddos.device
    rows2 = S * 2 * B
    r32 = _rows32(rows2)
    gf, bf = P["transformer_source.layer_norm.weight"], P["transformer_source.layer_norm.bias"]
    head = None
    if head_fused_bwd(H, cfg.T) and ddos.is_contiguous():
        # output layer + final LayerNorm backward inside the source encoder's first ffn_bwd launch
        dx = None
        head = (gf, bf, xhat_f, rstd_f, ddos, P["out_layer.weight"],
                ("transformer_source.layer_norm.weight", "transformer_source.layer_norm.bias", "out_layer.weight", "out_layer.bias"))
    else:
        pld = 3 * H + 1
        part = sink.scratch(r32, pld)
        dx = _empty(dev, rows2, H)
        ops.ln_rowdot_bwd(ddos, xhat_f, rstd_f, gf, bf, P["out_layer.weight"], dx, part, S, 2 * B, H)
        sink.add(part, 0, G["transformer_source.layer_norm.weight"], r32, pld, H)
        sink.add(part, H, G["transformer_source.layer_norm.bias"], r32, pld, H)
        sink.add(part, 2 * H, G["out_layer.weight"], r32, pld, H)
        sink.add(part, 3 * H, G["out_layer.bias"], r32, pld, 1)
    # key gradient of the two cross attentions (dense [nmax*B (+1 spare), H] layout): the first layer processed overwrites
    # every key row, so no zero fill; the spare row (dense slot of ghost nodes) is never read (dense_normalize_bwd's ghost_row)
    dkv = _empty(dev, nmax * B + 1, H)
    dhs = encoder_bwd(P, G, "transformer_source", c3, dx, dkv, sink, dkv_fresh=True, head=head)
    if sink.wside is not None:
        sink.flush_on_side()                  # (weight-gradient stream: this encoder's jobs run under the next one's backward)
    dkvs = _empty(dev, rows2, H)              # self-attention: every row is a key row, the first layer overwrites
    ddosin = encoder_bwd(P, G, "transformer_self", c2, dhs, dkvs, sink, dkv_fresh=True, kv_needed_next=True)
    late_flush = _LATE_SELF_FLUSH in ("1", "2") or (_LATE_SELF_FLUSH == "auto" and len(a_g.keep) <= 2)
    if sink.wside is not None and not late_flush:
        sink.flush_on_side()
    sink.join()          # dkvs is produced on the side stream
    dpre = _empty(dev, rows2, H)         # key-side LN backward + the LeakyReLU backward behind it, one launch
    Wfc, Wfp = P["fc.weight"], P["fc_prompt.weight"]
    # round 6: ... and the two heads' input-gradient products behind it in the SAME launch (csrc/heads.hip): three launch-bound
    # kernels between two encoders' backward as one
    # (measured, tools/exp/r6_run5.sh, three interleaved rounds: cfg2 1.0672 -> 1.0576 ms; Electron-DOS hidden 256 6.971 -> 7.009 ms -
    #  at 12864 rows x 512 k the 64-row-tile GEMMs re-use the weights better than 3216 16-row workgroups: hidden <= 128 only)
    heads_one = (_HEADS_BWD_ONE_LAUNCH and H <= _HEADS_BWD_MAX_H and ops.heads_bwd_supported(H) and ddosin.is_contiguous()
                 and dkvs.is_contiguous())
    dE1 = _empty(dev, S * B, H)
    if heads_one:
        ops.heads_bwd(S, B, H, dkvs, kvs, rstd_s, ddosin, dosin, 0.01, dpre, Wfc, Wfp, dE1)
    else:
        ops.rownorm_bwd_act(dkvs, kvs, rstd_s, ddosin, dosin, 0.01, dpre, rows2, H)
    map0, map1 = rowmap(d=B, m=2 * B, c=1, off=0), rowmap(d=B, m=2 * B, c=1, off=B)
    R = _empty(dev, 2 * B, H)            # sum over the energy axis of dpre (filled on the side stream below)
    # (forward factored; R is filled later, on the side stream: deferred jobs only - and not with the late-flush experiment,
    #  whose flush_on_side() below would launch the B-row jobs before the reduce_rows that writes R is even queued)
    r_jobs = None
    if len(a_g.keep) > 2 and "fc.weight" in G and "fc_prompt.weight" in G:
        # the same factoring for the weight gradients: the column blocks that multiply the per-crystal inputs are
        # (sum_s dpre[s, b]) (x) [graph_b (| prompt_b)] - B-row jobs on R - and only the E1 block keeps its S * B rows
        E1_, graph_ = a_g.keep[0], a_g.keep[1]
        Gfc, Gfp = G["fc.weight"], G["fc_prompt.weight"]
        _wgrad_linear(sink, G, "fc.weight", "fc.bias", S * B, H, seg(dpre, rmap=map0), [seg(E1_)], keep=(dpre, E1_), dst=Gfc[:, :H])
        _wgrad_linear(sink, G, "fc_prompt.weight", "fc_prompt.bias", S * B, H, seg(dpre, rmap=map1), [seg(E1_)], keep=(dpre,),
                      dst=Gfp[:, :H])

        def r_jobs():
            _wgrad_linear(sink, G, "fc.weight", None, B, H, seg(R[:B]), [seg(graph_)], keep=(R, graph_), dst=Gfc[:, H:])
            _wgrad_linear(sink, G, "fc_prompt.weight", None, B, H, seg(R[B:]), [seg(graph_), seg(prow)], keep=(R, prow), dst=Gfp[:, H:])
        if not late_flush:
            r_jobs()
            r_jobs = None
    else:
        _wgrad_linear(sink, G, "fc.weight", "fc.bias", S * B, H, seg(dpre, rmap=map0), a_g.segs, keep=(dpre,))
        _wgrad_linear(sink, G, "fc_prompt.weight", "fc_prompt.bias", S * B, H, seg(dpre, rmap=map1), a_s.segs, keep=(dpre,))
    if not heads_one:
        ops.gemm(S * B, H, [seg(dpre, rmap=map0)], Wfc[:, :H], dE1, w_layout=1)
        ops.gemm(S * B, H, [seg(dpre, rmap=map1)], Wfp[:, :H], dE1, w_layout=1, res=dE1)
    if sink.wside is not None and late_flush and _LATE_SELF_FLUSH != "2":
        # experiment: the self encoder's weight gradients (+ the two heads') start behind the small head kernels above
        # instead of in front of them (where the group's long-lived workgroups make these 8-15 us kernels wait)
        sink.flush_on_side()
    if r_jobs is not None:
        # (late flush: the B-row jobs read R, which the side stream fills below - they are described BEHIND that flush and go with
        #  the next one, which every mode places behind the join with the side stream)
        r_jobs()
    # graph / prompt inputs are constant over the energy axis: reduce over s first, then a [2B,H] GEMM.  Their
    # consumers (decoder backward, prompt-embedding gradient) come after the first encoder's backward: side stream.
    dgraph = _empty(dev, B, H)
    hp = H // 2
    dprow = _empty(dev, B, hp)

    def _const_inputs_bwd():
        ops.reduce_rows(dpre.data_ptr(), H, R.data_ptr(), H, 2 * B, S, 1, 2 * B, H)
        ops.gemm(B, H, [seg(R[:B])], Wfc[:, H:2 * H], dgraph, w_layout=1)
        ops.gemm(B, H, [seg(R[B:])], Wfp[:, H:2 * H], dgraph, w_layout=1, res=dgraph)
        ops.gemm(B, hp, [seg(R[B:])], Wfp[:, 2 * H:], dprow, w_layout=1)
        ops.embed_rows_bwd(dprow.data_ptr(), hp, sysidx, G[cfg.prompt_key], B, G[cfg.prompt_key].shape[0], hp)
        # the decoder's dgrad needs only dgraph: it runs here, off the main chain, and its pooled block is added to the
        # node gradient by the dense-key backward launch below (one launch on the main stream instead of three)
        box_d["dcat"] = decoder_dgrad(P, cfg, m, dec_segs, dgraph)
    box_d = {}
    sink.on_side(_const_inputs_bwd, (dpre, R, dgraph, dprow))
    # first encoder (queries = energy embeddings broadcast over the batch)
    dX1 = encoder_bwd(P, G, "transformer", c1, dE1, dkv, sink, kv_needed_next=True)
    if sink.wside is not None and _LATE_SELF_FLUSH == "2":
        sink.flush_on_side()             # (experiment: the self encoder's group together with the first encoder's, under the GNN backward)
    # Every gradient of the transformer stacks, the heads and the embeddings is complete (or queued on the side
    # stream) here; what follows only touches the GNN trunk's parameters.  mid_hook: data-parallel training reduces
    # and all-reduces that early bucket now, underneath the GNN backward (train.Trainer).
    sink.join()          # dkv (dense keys): the earlier layers' key-gradient kernels ran on the side stream
    # gradient of the energy embeddings (sum over the batch): nobody waits for it - side stream, under the GNN backward
    sink.on_side(lambda: ops.reduce_rows(dX1.data_ptr(), H, G["embeddings.weight"].data_ptr(), H, S, B, B, 1, H), (dX1,))
    late_hook = None
    if mid_hook is not None:          # (after the join: the early bucket's reduction must not sit between the main
        if _MID_HOOK_LATE and H <= _MID_HOOK_LATE_MAX_H and dx_ext is None and cfg.L >= 1:
            late_hook = mid_hook      # (experiment: behind the last layer's NodeModel backward launch, gnn_bwd)
        else:
            mid_hook(sink)            #  stream and the dk/dv kernels it is waiting for)
    # node embeddings: dense keys + pooled decoder input (+ external grad on the returned x)
    dxL = _empty(dev, N, H)
    dcat = box_d["dcat"]
    Kd = dec_segs.K
    def dense_launch():
        ops.dense_normalize_pool_bwd(dkv, kvhat, rstd_n, m.dense_row, dcat.data_ptr() + 4 * (Kd - H), Kd, m.node_graph, B, dxL, N, H,
                                     False, ghost_row=nmax * B)
    dx_pre = None
    if dx_ext is not None or cfg.L < 1:
        dense_launch()
    else:
        # (round 6: inside the last message-passing layer's NodeModel backward launch, when that runs column-split: gnn_bwd)
        dx_pre = (dict(kind="dense", dkv=dkv, kvhat=kvhat, rstd_nodes=rstd_n, dense_row=m.dense_row, dpool_ptr=dcat.data_ptr() + 4 * (Kd - H),
                       ld_dpool=Kd, node_graph=m.node_graph, num_graphs=B, ghost_row=nmax * B), dense_launch)
        sink._keep.extend([dkv, dcat])
    du_seg = decoder_bwd(P, G, cfg, m, dec_segs, dgraph, None, sink, dcat=dcat)
    if dx_ext is not None:
        dxL.add_(dx_ext)
    sink.after_first_node = late_hook
    gnn_trunk_bwd(P, G, cfg, m, ctrunk, dxL, du_seg, sink, dx_pre=dx_pre)
    sink.flush()


# ---- hidden > 256 (`utils.py:25-43` takes any --hidden) --------------------------------------------------------------------
# The MFMA attention kernels, the fused feed-forward / NodeModel kernels and the one-tile row epilogues stop at 256-wide
# rows.  Wider models run the SAME math through the unfused building blocks: the K != V encoder path (layer_norms[0] as
# three LayerNorm launches, scores / softmax / P.V from csrc/attention_kv.hip, the feed-forward half as two GEMMs) on the RAW
# dense keys (dosx_dense_slots = to_dense_batch alone), plain GEMMs + row kernels for the 2H-wide LayerNorms of the GNN
# blocks (mlp_ln_fwd / _bwd above).  One stream, no launch-saving tricks: a correctness path for rare shapes, pinned against
# the oracle like every other path (tests/test_gpu_models.py).  Limit: hidden <= 512 (LayerNorm prologues of dosx_gemm).
def _sum_rows(dev, terms, rows, H, keep):
    """sum of equally shaped [rows, H] tensors (ops.mask_residual: out = res + a)"""
    acc = terms[0]
    for t in terms[1:]:
        out = _empty(dev, rows, H)
        ops.mask_residual(t, None, acc, out, None, rows, H)
        keep.extend([t, acc])
        acc = out
    return acc


def _dostransformer_fwd_wide(P: Params, cfg: ModelCfg, g, m: GraphMeta, dr):
    H, S, T, B, N = cfg.H, cfg.S, cfg.T, m.num_graphs, m.num_nodes
    nmax = m.n_max
    dev = P["embeddings.weight"].device
    xL, u, ctrunk = gnn_trunk_fwd(P, cfg, g, m)
    dense = _empty(dev, nmax * B, H)
    ops.dense_slots(xL, m.graph_ptr, dense, B, nmax, H)
    graph, dec_segs = decoder_fwd(P, cfg, m, xL, u)
    sysidx = _i32(g.system)
    hp = H // 2
    prow = _empty(dev, B, hp)
    ops.embed_rows(P[cfg.prompt_key], sysidx, prow, B, hp)
    # the energy embeddings expanded over the batch (`embeddings.weight.unsqueeze(1).expand(S, B, H)`): row (s, b) = row s.
    # The index is a constant of the (S, B) shape: built once by torch, also valid when the program is replayed.
    sidx = torch.arange(S, device=dev, dtype=torch.int32).repeat_interleave(B).contiguous()
    x0 = _empty(dev, S * B, H)
    ops.embed_rows(P["embeddings.weight"], sidx, x0, S * B, H)
    E1, c1 = encoder_kv_fwd(P, "transformer", x0, dense, dense, S, B, nmax, B, H, T, drop=dr(0))
    dosin = _empty(dev, S * 2 * B, H)
    modB = rowmap(d=B, m=0, c=1)
    a_g = SegList([seg(E1), seg(graph, rmap=modB)], [E1, graph])
    a_s = SegList([seg(E1), seg(graph, rmap=modB), seg(prow, rmap=modB)], [E1, graph, prow])
    ops.gemm(S * B, H, a_g.segs, P["fc.weight"], dosin, bias=P["fc.bias"], act=ACT_LEAKY, act_slope=0.01,
             out_map=rowmap(d=B, m=2 * B, c=1, off=0))
    ops.gemm(S * B, H, a_s.segs, P["fc_prompt.weight"], dosin, bias=P["fc_prompt.bias"], act=ACT_LEAKY, act_slope=0.01,
             out_map=rowmap(d=B, m=2 * B, c=1, off=B))
    hs, c2 = encoder_kv_fwd(P, "transformer_self", dosin, dosin, dosin, S, 2 * B, S, 2 * B, H, T, drop=dr(64))
    hsrc, c3 = encoder_kv_fwd(P, "transformer_source", hs, dense, dense, S, 2 * B, nmax, B, H, T, final_ln=False, drop=dr(128))
    xhat_f, rstd_f, dos = _empty(dev, S * 2 * B, H), _empty(dev, S * 2 * B), _empty(dev, 2 * B, S)
    ops.ln_rowdot(hsrc, P["transformer_source.layer_norm.weight"], P["transformer_source.layer_norm.bias"],
                  P["out_layer.weight"], P["out_layer.bias"], xhat_f, rstd_f, dos, S, 2 * B, H)
    ctx = ("wide", ctrunk, c1, dec_segs, sysidx, sidx, prow, dosin, a_g, a_s, c2, c3, xhat_f, rstd_f, xL, x0, dense, hsrc)
    return dos, xL, ctx


def _dostransformer_bwd_wide(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, ctx, ddos: torch.Tensor,
                             dx_ext: Optional[torch.Tensor], sink: GradSink, mid_hook=None) -> None:
    (_, ctrunk, c1, dec_segs, sysidx, sidx, prow, dosin, a_g, a_s, c2, c3, xhat_f, rstd_f, xL, x0, dense, hsrc) = ctx
    H, S, B, N = cfg.H, cfg.S, m.num_graphs, m.num_nodes
    nmax = m.n_max
    dev = ddos.device
    rows2 = S * 2 * B
    r32 = _rows32(rows2)
    keep = sink._keep
    pld = 3 * H + 1
    part = sink.scratch(r32, pld)
    dx = _empty(dev, rows2, H)
    ops.ln_rowdot_bwd(ddos, xhat_f, rstd_f, P["transformer_source.layer_norm.weight"], P["transformer_source.layer_norm.bias"],
                      P["out_layer.weight"], dx, part, S, 2 * B, H)
    sink.add(part, 0, G["transformer_source.layer_norm.weight"], r32, pld, H)
    sink.add(part, H, G["transformer_source.layer_norm.bias"], r32, pld, H)
    sink.add(part, 2 * H, G["out_layer.weight"], r32, pld, H)
    sink.add(part, 3 * H, G["out_layer.bias"], r32, pld, 1)
    dhs, dk3, dv3 = encoder_kv_bwd(P, G, "transformer_source", c3, dx, sink)
    ddosin, dk2, dv2 = encoder_kv_bwd(P, G, "transformer_self", c2, dhs, sink)
    dsum = _sum_rows(dev, [ddosin, dk2, dv2], rows2, H, keep)        # dosin is query, key and value of the self encoder
    dpre = _empty(dev, rows2, H)
    ops.act_bwd(dsum, dosin, 0.01, dpre)                             # F.leaky_relu behind fc / fc_prompt
    keep.extend([dx, dhs, dsum])
    map0, map1 = rowmap(d=B, m=2 * B, c=1, off=0), rowmap(d=B, m=2 * B, c=1, off=B)
    _wgrad_linear(sink, G, "fc.weight", "fc.bias", S * B, H, seg(dpre, rmap=map0), a_g.segs, keep=(dpre,))
    _wgrad_linear(sink, G, "fc_prompt.weight", "fc_prompt.bias", S * B, H, seg(dpre, rmap=map1), a_s.segs, keep=(dpre,))
    Wfc, Wfp = P["fc.weight"], P["fc_prompt.weight"]
    dE1 = _empty(dev, S * B, H)
    ops.gemm(S * B, H, [seg(dpre, rmap=map0)], Wfc[:, :H], dE1, w_layout=1)
    ops.gemm(S * B, H, [seg(dpre, rmap=map1)], Wfp[:, :H], dE1, w_layout=1, res=dE1)
    R = _empty(dev, 2 * B, H)                                        # graph / prompt inputs are constant over the energy axis
    dgraph = _empty(dev, B, H)
    hp = H // 2
    dprow = _empty(dev, B, hp)
    ops.reduce_rows(dpre.data_ptr(), H, R.data_ptr(), H, 2 * B, S, 1, 2 * B, H)
    ops.gemm(B, H, [seg(R[:B])], Wfc[:, H:2 * H], dgraph, w_layout=1)
    ops.gemm(B, H, [seg(R[B:])], Wfp[:, H:2 * H], dgraph, w_layout=1, res=dgraph)
    ops.gemm(B, hp, [seg(R[B:])], Wfp[:, 2 * H:], dprow, w_layout=1)
    ops.embed_rows_bwd(dprow.data_ptr(), hp, sysidx, G[cfg.prompt_key], B, G[cfg.prompt_key].shape[0], hp)
    keep.extend([R, dprow])
    dX1, dk1, dv1 = encoder_kv_bwd(P, G, "transformer", c1, dE1, sink)
    ops.reduce_rows(dX1.data_ptr(), H, G["embeddings.weight"].data_ptr(), H, S, B, B, 1, H)       # sum over the batch
    keep.extend([dX1, dE1])
    if mid_hook is not None:
        mid_hook(sink)
    ddense = _sum_rows(dev, [dk1, dv1, dk3, dv3], nmax * B, H, keep)
    dxL = _empty(dev, N, H)
    ops.dense_slots_bwd(ddense, m.dense_row, dxL, N, H, False, ghost_row=nmax * B)
    keep.append(ddense)
    du_seg = decoder_bwd(P, G, cfg, m, dec_segs, dgraph, dxL, sink)
    if dx_ext is not None:
        dxL.add_(dx_ext)
    gnn_trunk_bwd(P, G, cfg, m, ctrunk, dxL, du_seg, sink)
    sink.flush()


# ---- GNN-only variants (graphnetwork_phonon.py:48-72, graphnetwork.py:26-43) -----------------------
def graphnetwork_fwd(P: Params, cfg: ModelCfg, g, m: GraphMeta):
    H, S, B = cfg.H, cfg.S, m.num_graphs
    dev = P["embeddings.weight"].device
    expected = 118 if cfg.kind == "phonon" else 200          # graphnetwork_phonon.py:150-153 / graphnetwork.py:96-99
    node_key = "GN_encoder.node_encoder" if g.x.shape[1] == expected else "GN_encoder.node_encoder_prompt"
    xL, u, ctrunk = gnn_trunk_fwd(P, cfg, g, m, node_key)
    graph, dec_segs = decoder_fwd(P, cfg, m, xL, u)
    emb = P["embeddings.weight"]
    a = SegList([seg(emb, rmap=rowmap(d=B, m=1, c=0)), seg(graph, rmap=rowmap(d=B, m=0, c=1))], [emb, graph])
    hid = _empty(dev, S * B, H)
    ops.gemm(S * B, H, a.segs, P["out_layer.0.weight"], hid, bias=P["out_layer.0.bias"], act=ACT_LEAKY, act_slope=0.01)
    dos = _empty(dev, B, S)
    ops.rowdot(hid, P["out_layer.2.weight"], P["out_layer.2.bias"], dos, S, B, H)
    return dos, xL, (ctrunk, dec_segs, a, hid, xL)


def graphnetwork_bwd(P: Params, G: Params, cfg: ModelCfg, m: GraphMeta, ctx, ddos: torch.Tensor,
                     dx_ext: Optional[torch.Tensor], sink: GradSink) -> None:
    ctrunk, dec_segs, a, hid, xL = ctx
    H, S, B, N = cfg.H, cfg.S, m.num_graphs, m.num_nodes
    dev = ddos.device
    rows = S * B
    r32 = _rows32(rows)
    part = sink.scratch(r32, H + 1)
    dhid = _empty(dev, rows, H)
    ops.rowdot_bwd(ddos, hid, P["out_layer.2.weight"], dhid, part, S, B, H)
    sink.add(part, 0, G["out_layer.2.weight"], r32, H + 1, H)
    sink.add(part, H, G["out_layer.2.bias"], r32, H + 1, 1)
    dpre = _empty(dev, rows, H)
    ops.act_bwd(dhid, hid, 0.01, dpre)
    _wgrad_linear(sink, G, "out_layer.0.weight", "out_layer.0.bias", rows, H, seg(dpre), a.segs, keep=(dpre,))
    W0 = P["out_layer.0.weight"]
    Rs = _empty(dev, S, H)
    ops.reduce_rows(dpre.data_ptr(), H, Rs.data_ptr(), H, S, B, B, 1, H)          # sum over the batch
    ops.gemm(S, H, [seg(Rs)], W0[:, :H], G["embeddings.weight"], w_layout=1)
    Rb = _empty(dev, B, H)
    ops.reduce_rows(dpre.data_ptr(), H, Rb.data_ptr(), H, B, S, 1, B, H)          # sum over the energy bins
    dgraph = _empty(dev, B, H)
    ops.gemm(B, H, [seg(Rb)], W0[:, H:], dgraph, w_layout=1)
    dxL = ops.zeros(dev, N, H)
    du_seg = decoder_bwd(P, G, cfg, m, dec_segs, dgraph, dxL, sink)
    if dx_ext is not None:
        dxL.add_(dx_ext)
    gnn_trunk_bwd(P, G, cfg, m, ctrunk, dxL, du_seg, sink)
    sink.flush()
