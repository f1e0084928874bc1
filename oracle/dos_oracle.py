"""ORACLE — test infrastructure, NOT product code.

CPU restatement (pure torch, functional, autograd-differentiable) of the
DOSTransformer forward / loss / AdamW hot path, written from the reference's
behaviour.  Every function cites the reference file:line it follows (paths are
relative to the upstream repo root).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``dostransformer_amd`` never does.

Parity pin: the reference repository contains NO tests, golden vectors or
fixtures of its own, so this oracle is pinned against outputs of
the reference itself, generated in the build container by importing
``/root/reference`` (see ``tests/golden/make_golden.py``; vectors committed under
``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).

Third-party arithmetic that is NOT in the reference tree and whose versions the
reference does not pin (no requirements file): ``torch_scatter.scatter_sum /
scatter_mean``, ``torch_geometric.utils.to_dense_batch``,
``e3nn.o3.spherical_harmonics(l<=1, normalize=True, normalization='component')``,
``e3nn.nn.models.gate_points_2101.smooth_cutoff``.  Their published semantics are
restated below (``scatter_sum`` .. ``smooth_cutoff``); parity AT THAT BOUNDARY is
therefore "unpinned" (same restatement is used as the stand-in when the fixtures
are generated), everything above it is the reference's own code.

Parameters are passed as a ``dict`` keyed exactly like the reference modules'
``state_dict()``.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# third-party semantics (not under /root/reference; versions unpinned upstream)
# --------------------------------------------------------------------------------------
def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_scatter.scatter_sum(src, index, dim=0, dim_size): index_add into zeros."""
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.index_add(0, index, src)


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """torch_scatter.scatter_mean: sum / per-index count clamped to >= 1."""
    s = scatter_sum(src, index, dim_size)
    cnt = torch.bincount(index, minlength=dim_size).clamp(min=1).to(src.dtype)
    return s / cnt.reshape((-1,) + (1,) * (src.dim() - 1))


def to_dense_batch(x: torch.Tensor, batch: torch.Tensor, num_graphs: int, n_max: int) -> torch.Tensor:
    """torch_geometric.utils.to_dense_batch: zero padded [B, Nmax, F], node order kept."""
    counts = torch.bincount(batch, minlength=num_graphs)
    ptr = torch.zeros(num_graphs + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(counts, 0)
    pos = torch.arange(x.shape[0]) - ptr[batch]
    out = x.new_zeros(num_graphs * n_max, x.shape[1])
    out = out.index_copy(0, batch * n_max + pos, x)
    return out.reshape(num_graphs, n_max, x.shape[1])


def smooth_cutoff(x: torch.Tensor) -> torch.Tensor:
    """e3nn gate_points_2101.smooth_cutoff: u=2(x-1); (1-cos(pi u))/2, 0 if u>0, 1 if u<-1."""
    u = 2 * (x - 1)
    y = (1 - torch.cos(math.pi * u)) / 2
    y = torch.where(u > 0, torch.zeros_like(y), y)
    y = torch.where(u < -1, torch.ones_like(y), y)
    return y


def spherical_harmonics_l1(vec: torch.Tensor) -> torch.Tensor:
    """e3nn o3.spherical_harmonics('1x0e+1x1o', vec, normalize=True, 'component'):
    [1, sqrt3*x^, sqrt3*y^, sqrt3*z^] with x^ = vec / max(|vec|, 1e-12) (F.normalize)."""
    unit = F.normalize(vec, dim=-1)
    return torch.cat([torch.ones_like(vec[:, :1]), math.sqrt(3.0) * unit], dim=1)


# --------------------------------------------------------------------------------------
# small torch.nn restatements
# --------------------------------------------------------------------------------------
def _linear(p: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, p[key + ".weight"], p[key + ".bias"])


def _prelu(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return torch.where(x >= 0, x, w * x)


def _layer_norm(p: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), p[key + ".weight"], p[key + ".bias"], 1e-5)


def _mlp_prelu(p: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(Linear, PReLU, Linear) — the Encoder MLPs, DOSTransformer_phonon.py:129-130."""
    h = _prelu(_linear(p, key + ".0", x), p[key + ".1.weight"])
    return _linear(p, key + ".2", h)


def _mlp_ln_prelu(p: Params, key: str, x: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(Linear, LayerNorm, PReLU, Linear) — Edge/Node MLPs, DOSTransformer_phonon.py:193,204."""
    h = _layer_norm(p, key + ".1", _linear(p, key + ".0", x))
    h = _prelu(h, p[key + ".2.weight"])
    return _linear(p, key + ".3", h)


# --------------------------------------------------------------------------------------
# a1: edge features (phonon)        DOSTransformer_phonon.py:74-77, graphnetwork_phonon.py:53-56
# --------------------------------------------------------------------------------------
def edge_features_sh1(edge_vec: torch.Tensor, r_max: float = 4.0) -> torch.Tensor:
    sh = spherical_harmonics_l1(edge_vec)
    length = edge_vec.norm(dim=1)
    return smooth_cutoff(length / r_max)[:, None] * sh


# --------------------------------------------------------------------------------------
# a3-a6: message passing stack       DOSTransformer_phonon.py:81-84,148-171,190-212
#                                    DOSTransformer.py:56-59,125-148,168-190
# --------------------------------------------------------------------------------------
def processor(p: Params, prefix: str, x, edge_index, e, mean: bool):
    row, col = edge_index[0], edge_index[1]
    e_out = _mlp_ln_prelu(p, prefix + ".edge_model.edge_mlp", torch.cat([x[row], x[col], e], 1))
    agg = (scatter_mean if mean else scatter_sum)(e_out, col, x.shape[0])
    x_out = _mlp_ln_prelu(p, prefix + ".node_model.node_mlp_2", torch.cat([x, agg], 1))
    return x_out, e_out


def gnn_stack(p: Params, x, edge_index, e, n_layers: int, mean: bool):
    for l in range(n_layers):
        dx, de = processor(p, f"stacked_processor.{l}", x, edge_index, e, mean)
        x = x + dx
        e = e + de
    return x, e


# --------------------------------------------------------------------------------------
# a9-a11: attention / transformer encoder       layers/multihead_attention.py:49-76,
#                                               layers/transformer.py:46-79,120-157
# --------------------------------------------------------------------------------------
def multihead_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, drop_mask=None) -> torch.Tensor:
    """(seq,batch,dim) in/out. No projections, no mask, no head split; softmax in fp32
    (`multihead_attention.py:68-74`); scaling = embed_dim**-0.5 (`:20`).
    drop_mask [batch, tgt, src]: `F.dropout(attn_weights, p, training)` (`:70`) with the Bernoulli draw made explicit - the
    multiplier M in {0, 1/(1-p)} (None = eval mode / p = 0)."""
    dim = q.shape[2]
    w = torch.bmm(q.transpose(0, 1), k.transpose(0, 1).transpose(1, 2)) * (dim ** -0.5)
    w = F.softmax(w.float(), dim=-1).type_as(w)
    if drop_mask is not None:
        w = w * drop_mask.to(w.dtype)
    return torch.bmm(w, v.transpose(0, 1)).transpose(0, 1)


def encoder_layer(p: Params, prefix: str, x, x_k, x_v, drop_mask=None):
    """Pre-norm block; layer_norms.0 is shared by q, k and v (`transformer.py:131-134`).
    drop_mask: None, the attention-dropout multiplier [batch, tgt, src], or a dict of explicit multipliers for the layer's
    dropout sites - "attn" (`multihead_attention.py:70`), "res1" (`transformer.py:137`, on the attention output), "relu"
    (`:145`, on relu(fc1)) and "res2" (`:147`, on the fc2 output); missing keys = no dropout there."""
    dm = drop_mask if isinstance(drop_mask, dict) else {"attn": drop_mask}
    mul = lambda t, key: t if dm.get(key) is None else t * dm[key].to(t.dtype).reshape(t.shape)
    r = x
    q = _layer_norm(p, prefix + ".layer_norms.0", x)
    k = _layer_norm(p, prefix + ".layer_norms.0", x_k)
    v = _layer_norm(p, prefix + ".layer_norms.0", x_v)
    x = r + mul(multihead_attention(q, k, v, dm.get("attn")), "res1")
    r = x
    y = _layer_norm(p, prefix + ".layer_norms.1", x)
    y = _linear(p, prefix + ".fc2", mul(F.relu(_linear(p, prefix + ".fc1", y)), "relu"))
    return r + mul(y, "res2")


def transformer_encoder(p: Params, prefix: str, x, x_k, x_v, n_layers: int, drop_masks=None):
    """x_k / x_v are NOT updated between layers (`transformer.py:72-73`).  drop_masks: one attention-dropout multiplier
    per layer (training mode with attn_dropout > 0), see multihead_attention."""
    for t in range(n_layers):
        x = encoder_layer(p, f"{prefix}.layers.{t}", x, x_k, x_v, None if drop_masks is None else drop_masks[t])
    return _layer_norm(p, prefix + ".layer_norm", x)


# --------------------------------------------------------------------------------------
# a12: full models
# --------------------------------------------------------------------------------------
def _batch_info(g) -> Tuple[int, int]:
    nb = int(g.system.shape[0]) if "system" in g else int(g.batch.max()) + 1
    counts = torch.bincount(g.batch, minlength=nb)
    m = getattr(g, "meta", None)
    n_max = m.n_max if m is not None else int(counts.max())
    return nb, n_max


def _heads(p: Params, energies, graph, x_dense, prompt_rows, n_t: int, drop_masks=None):
    """Shared tail of both DOSTransformer variants (`DOSTransformer_phonon.py:93-117`,
    `DOSTransformer.py:68-91`): global branch, then prompt ('system') branch.
    drop_masks (training mode, attn_drop > 0): {"transformer_self" / "transformer_source": [per layer [2B, S, Nk]]} with
    the global branch's draws in rows [0,B) and the system branch's in [B,2B)."""
    outs = []
    nb = energies.shape[1]
    dm = lambda key, branch: None if drop_masks is None else [m[branch * nb:(branch + 1) * nb] for m in drop_masks[key]]
    for branch in (0, 1):
        if branch == 0:
            h = F.leaky_relu(_linear(p, "fc", torch.cat([energies, graph], 2)))
        else:
            h = F.leaky_relu(_linear(p, "fc_prompt", torch.cat([energies, graph, prompt_rows], 2)))
        h = transformer_encoder(p, "transformer_self", h, h, h, n_t, dm("transformer_self", branch))
        h = transformer_encoder(p, "transformer_source", h, x_dense, x_dense, n_t, dm("transformer_source", branch))
        outs.append(_linear(p, "out_layer", h).squeeze(2).T)
    return outs[0], outs[1]


def dostransformer_phonon_forward(p: Params, g, n_layers: int, n_t: int, drop_masks=None):
    """`embedder_phDOS/DOSTransformer_phonon.py:66-119` -> (dos_global [B,51], x [N,H], dos_system [B,51])."""
    nb, n_max = _batch_info(g)
    s = p["embeddings.weight"].shape[0]
    energies = p["embeddings.weight"]                                  # :71 (ids = arange)
    e = edge_features_sh1(g.edge_vec)                                   # :74-77
    x = _mlp_prelu(p, "GN_encoder.node_encoder", g.x)                  # :141
    e = _mlp_prelu(p, "GN_encoder.edge_encoder", e)                    # :142
    energies = energies[:, None, :].expand(s, nb, energies.shape[1])   # :143
    x, e = gnn_stack(p, x, g.edge_index, e, n_layers, mean=True)       # :81-84
    x_dense = to_dense_batch(x, g.batch, nb, n_max).transpose(0, 1)    # :86-87
    energies = transformer_encoder(p, "transformer", energies, x_dense, x_dense, n_t,
                                   None if drop_masks is None else drop_masks["transformer"])   # :88
    graph = _linear(p, "GN_decoder.mlp.0", scatter_sum(x, g.batch, nb))               # :90, :180-181
    graph = graph[None].expand(s, nb, graph.shape[1])                  # :91
    prompt_rows = p["prompt_token.weight"][g.system][None].expand(s, nb, -1)            # :105
    dos_global, dos_system = _heads(p, energies, graph, x_dense, prompt_rows, n_t, drop_masks)
    return dos_global, x, dos_system


def dostransformer_forward(p: Params, g, n_layers: int, n_t: int, drop_masks=None):
    """`embedder_eDOS/DOSTransformer.py:45-93` (note the upstream spelling ``promt_token``)."""
    nb, n_max = _batch_info(g)
    s = p["embeddings.weight"].shape[0]
    energies = p["embeddings.weight"]
    x = _mlp_prelu(p, "GN_encoder.node_encoder", g.x)                  # :116
    e = _mlp_prelu(p, "GN_encoder.edge_encoder", g.edge_attr)          # :117
    energies = energies[:, None, :].expand(s, nb, energies.shape[1])   # :118
    u = _mlp_prelu(p, "GN_encoder.global_encoder", g.glob.reshape(-1, 2))   # :119-120
    x, e = gnn_stack(p, x, g.edge_index, e, n_layers, mean=False)      # :56-59, sum aggregation :187
    x_dense = to_dense_batch(x, g.batch, nb, n_max).transpose(0, 1)    # :61-62
    energies = transformer_encoder(p, "transformer", energies, x_dense, x_dense, n_t,
                                   None if drop_masks is None else drop_masks["transformer"])   # :63
    graph = _linear(p, "GN_decoder.mlp.0", torch.cat([u, scatter_sum(x, g.batch, nb)], 1))  # :158-159
    graph = graph[None].expand(s, nb, graph.shape[1])                  # :65 (.repeat)
    prompt_rows = p["promt_token.weight"][g.system][None].expand(s, nb, -1)             # :79
    dos_global, dos_system = _heads(p, energies, graph, x_dense, prompt_rows, n_t, drop_masks)
    return dos_global, x, dos_system


# --------------------------------------------------------------------------------------
# a13: GNN-only variants
# --------------------------------------------------------------------------------------
def _gn_head(p: Params, energies, graph):
    h = _linear(p, "out_layer.0", torch.cat([energies, graph], 2))
    return _linear(p, "out_layer.2", F.leaky_relu(h)).squeeze(2).T


def graphnetwork_phonon_forward(p: Params, g, n_layers: int):
    """`embedder_phDOS/graphnetwork_phonon.py:48-72` -> dos [B,51]."""
    nb, _ = _batch_info(g)
    s = p["embeddings.weight"].shape[0]
    e = edge_features_sh1(g.edge_vec)
    enc = "GN_encoder.node_encoder" if g.x.shape[1] == 118 else "GN_encoder.node_encoder_prompt"   # :150-153
    x = _mlp_prelu(p, enc, g.x)
    e = _mlp_prelu(p, "GN_encoder.edge_encoder", e)
    energies = p["embeddings.weight"][:, None, :].expand(s, nb, -1)
    x, e = gnn_stack(p, x, g.edge_index, e, n_layers, mean=True)
    graph = _linear(p, "GN_decoder.mlp.0", scatter_sum(x, g.batch, nb))
    return _gn_head(p, energies, graph[None].expand(s, nb, -1))


def graphnetwork_forward(p: Params, g, n_layers: int):
    """`embedder_eDOS/graphnetwork.py:26-43` -> (dos [B,201], x)."""
    nb, _ = _batch_info(g)
    s = p["embeddings.weight"].shape[0]
    enc = "GN_encoder.node_encoder" if g.x.shape[1] == 200 else "GN_encoder.node_encoder_prompt"   # :96-99
    x = _mlp_prelu(p, enc, g.x)
    e = _mlp_prelu(p, "GN_encoder.edge_encoder", g.edge_attr)
    energies = p["embeddings.weight"][:, None, :].expand(s, nb, -1)
    u = _mlp_prelu(p, "GN_encoder.global_encoder", g.glob.reshape(-1, 2))
    x, e = gnn_stack(p, x, g.edge_index, e, n_layers, mean=False)
    graph = _linear(p, "GN_decoder.mlp.0", torch.cat([u, scatter_sum(x, g.batch, nb)], 1))
    return _gn_head(p, energies, graph[None].expand(s, nb, -1)), x


# --------------------------------------------------------------------------------------
# a14: losses and optimiser step of the callers
# --------------------------------------------------------------------------------------
def loss_phonon(dos_global, dos_system, phdos, beta: float = 1.0):
    """`main_phDOS.py:109-114`: ONE rmse over all B*51 elements per branch."""
    return torch.sqrt(F.mse_loss(dos_global, phdos)) + beta * torch.sqrt(F.mse_loss(dos_system, phdos))


def loss_edos(dos_global, dos_system, y_ft, beta: float = 1.0):
    """`main_eDOS.py:111-123`: clamp target at 0, mean over crystals of per-crystal rmse."""
    y = torch.where(y_ft < 0, torch.zeros_like(y_ft), y_ft).reshape(dos_global.shape[0], -1)
    rg = torch.sqrt(((y - dos_global) ** 2).mean(dim=1)).mean()
    rs = torch.sqrt(((y - dos_system) ** 2).mean(dim=1)).mean()
    return rg + beta * rs


def adamw_step(params: Params, grads: Dict[str, torch.Tensor], state: Dict[str, dict], lr: float,
               weight_decay: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.AdamW as the callers configure it (`main_eDOS.py:93`, `main_phDOS.py:92`):
    decoupled decay, bias-corrected moments; params whose grad is None are skipped
    entirely (no decay either)."""
    b1, b2 = betas
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            st = state.setdefault(k, {"step": 0, "m": torch.zeros_like(params[k]), "v": torch.zeros_like(params[k])})
            st["step"] += 1
            t = st["step"]
            params[k].mul_(1 - lr * weight_decay)
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (st["v"].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
            params[k].addcdiv_(st["m"], denom, value=-lr / (1 - b1 ** t))


def train_step(kind: str, params: Params, state: Dict[str, dict], g, n_layers: int, n_t: int,
               lr: float = 1e-4, beta: float = 1.0):
    """One full training step of the reference callers on CPU; returns (loss, grads)."""
    leaves = {k: v.detach().requires_grad_(True) for k, v in params.items() if v.is_floating_point() and k != "version"
              and not k.endswith(".version")}
    if kind == "phonon":
        dg, _, ds = dostransformer_phonon_forward(leaves, g, n_layers, n_t)
        loss = loss_phonon(dg, ds, g.phdos, beta)
    elif kind == "edos":
        dg, _, ds = dostransformer_forward(leaves, g, n_layers, n_t)
        loss = loss_edos(dg, ds, g.y_ft, beta)
    else:
        raise ValueError(kind)
    names = list(leaves)
    gr = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gr))
    adamw_step(params, grads, state, lr)
    return loss.detach(), grads


# ---------------------------------------------------------------------------------------------------
# Evaluation loops: `utils.py:61-143`
# ---------------------------------------------------------------------------------------------------
def r2(y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
    """`utils.py:20-23`: sklearn `r2_score(y_true.flatten(), y_pred.flatten(), 'variance_weighted')`; on the
    flattened (single-output) arrays that is 1 - SS_res / SS_tot."""
    t, p = y_true.double().flatten(), y_pred.double().flatten()
    return float(1.0 - ((t - p) ** 2).sum() / ((t - t.mean()) ** 2).sum())


def _batch_metrics(y: torch.Tensor, pred: torch.Tensor):
    mse = ((y - pred) ** 2).mean(dim=1)
    return float(torch.sqrt(mse).mean()), float(mse.mean()), float((pred - y).abs().mean()), r2(y, pred)


def eval_phonon(p: Params, loader, n_layers: int, n_t: int):
    """`utils.py:117-143` (`test_phonon`): per-batch RMSE / MSE / MAE / R^2 of `preds_system`, averaged over batches."""
    acc = [0.0, 0.0, 0.0, 0.0]
    with torch.no_grad():
        for g in loader:
            dg, _, ds = dostransformer_phonon_forward(p, g, n_layers, n_t)
            y = g.phdos.reshape(dg.shape[0], -1)
            for i, v in enumerate(_batch_metrics(y, ds)):
                acc[i] += v
    return tuple(a / len(loader) for a in acc)


def eval_edos(p: Params, loader, n_layers: int, n_t: int):
    """`utils.py:61-112` (`test`): target and prediction clamped at 0 (`:76-78`), metrics as above, plus the
    concatenated (mp_id, preds, y, sum-pooled node embeddings) of `:90-110`."""
    acc = [0.0, 0.0, 0.0, 0.0]
    ids, preds, ys, embs = [], [], [], []
    with torch.no_grad():
        for g in loader:
            _, x_nodes, ds = dostransformer_forward(p, g, n_layers, n_t)
            y = torch.clamp(g.y_ft, min=0.0).reshape(len(g.mp_id), -1)
            ds = torch.clamp(ds, min=0.0)
            for i, v in enumerate(_batch_metrics(y, ds)):
                acc[i] += v
            ids += list(g.mp_id)
            preds.append(ds)
            ys.append(y)
            embs.append(scatter_sum(x_nodes, g.batch, len(g.mp_id)))
    return tuple(a / len(loader) for a in acc), (ids, torch.cat(preds), torch.cat(ys), torch.cat(embs))


# ---------------------------------------------------------------------------------------------------
# §8f-3 phonon featurisation: periodic neighbour list (`utils.py:267` calls ASE's neighbor_list("ijS", a, cutoff,
# self_interaction=True); ASE is not in the reference tree and has no pinned version -> PARITY UNPINNED for this
# function: it restates ASE's documented contract — all (i, j, S) with |pos[j]-pos[i]+S@cell| < cutoff, the (i,i,0)
# pair only with self_interaction — by brute force, and is pinned only by the crystallographic known answers in
# tests/test_oracle_golden.py (coordination shells of sc / fcc / bcc / hcp lattices).
def neighbor_list_bruteforce(pos, cell, cutoff, self_interaction=True):
    """numpy, one crystal.  Returns (i, j, S [E,3] int, D [E,3]) sorted by (i, j, S); D is summed in the order of
    `utils.py:271-273`: (pos[j]-pos[i]) + ((S0*a0 + S1*a1) + S2*a2)."""
    import numpy as np
    pos = np.asarray(pos, np.float64).reshape(-1, 3)
    cell = np.asarray(cell, np.float64).reshape(3, 3)
    n = pos.shape[0]
    inv = np.linalg.inv(cell)
    frac = pos @ inv
    spread = np.ceil(np.abs(frac[:, None, :] - frac[None, :, :]).max(axis=(0, 1))).astype(int) if n else np.zeros(3, int)
    reach = np.ceil(cutoff * np.linalg.norm(inv, axis=0)).astype(int) + spread + 1     # generous, not minimal
    ax = [np.arange(-r, r + 1) for r in reach]
    S = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3)                   # lexicographic
    sh = (S[:, 0:1] * cell[0][None, :] + S[:, 1:2] * cell[1][None, :]) + S[:, 2:3] * cell[2][None, :]
    ii, jj, ss, dd = [], [], [], []
    for i in range(n):
        for j in range(n):
            d = (pos[j] - pos[i])[None, :] + sh
            r2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            keep = r2 < cutoff * cutoff
            if i == j and not self_interaction:
                keep &= ~((S == 0).all(axis=1))
            k = np.nonzero(keep)[0]
            ii.append(np.full(k.shape, i)); jj.append(np.full(k.shape, j)); ss.append(S[k]); dd.append(d[k])
    if not ii:
        return np.zeros(0, int), np.zeros(0, int), np.zeros((0, 3), int), np.zeros((0, 3))
    return np.concatenate(ii), np.concatenate(jj), np.concatenate(ss), np.concatenate(dd)
