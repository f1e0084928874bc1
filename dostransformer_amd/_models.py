"""Shared implementation of the four drop-in model classes (phonon / eDOS x DOSTransformer /
Graphnetwork).  The public modules under ``embedder_phDOS`` / ``embedder_eDOS`` only fix the
constructor signatures and parameter creation order of their reference counterparts."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import functional as Fn
from ._fused import FusedModel


_SEED_MOD = 2 ** 62


def rank_seed_offset() -> int:
    """What a data-parallel rank adds to the (rank-independent) dropout base seed: ranks seeded alike must not draw the
    same masks for their different shards.  0 outside torch.distributed."""
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        return (0x9E3779B97F4A7C15 * (td.get_rank() + 1)) % _SEED_MOD
    return 0


def dos_device(P):
    return P["embeddings.weight"].device


class DOSTransformerBase(FusedModel):
    """forward(g) -> (dos_global [B,S], x [N,H], dos_system [B,S])   (`DOSTransformer_phonon.py:66-119`)."""
    _cfg: Fn.ModelCfg

    def _check_train_flags(self):
        pass          # (kept for callers of round 1: attention dropout is implemented now)

    # ---- attention dropout (`utils.py:40` --attn_drop -> TransformerEncoder(attn_dropout=...), multihead_attention.py:70)
    def _dropout(self, device, bump: bool):
        """None in eval mode / p = 0, else (p, seed_dev).  The seed lives in a device scalar so that a recorded program
        draws fresh masks on every replay; it starts from torch's RNG (follows torch.manual_seed) and is bumped once per
        forward pass (``bump``: the autograd path; train.Trainer bumps it itself, outside the recorded program)."""
        p = float(getattr(self, "_attn_drop", 0.0) or 0.0)
        if not self.training or p <= 0.0:
            return None
        seed = getattr(self, "_drop_seed", None)
        if seed is None or seed.device != device:
            val = (int(torch.randint(0, _SEED_MOD, (1,), dtype=torch.int64).item()) + rank_seed_offset()) % _SEED_MOD
            seed = torch.tensor([val], dtype=torch.int64).to(device)
            object.__setattr__(self, "_drop_seed", seed)
        elif bump:
            seed.add_(1)
        return p, seed

    def _program_fwd(self, P, g, m, bump_seed: bool = True, per_crystal_keys: bool = False):
        dos, xL, ctx = Fn.dostransformer_fwd(P, self._cfg, g, m, drop=self._dropout(dos_device(P), bump_seed),
                                             per_crystal_keys=per_crystal_keys)
        B = m.num_graphs
        return dos[:B], xL, dos[B:], (ctx, dos)

    def _program_bwd(self, P, G, m, saved, grads, sink):
        ctx, dos = saved
        dg, dx, ds = grads
        B = m.num_graphs
        ddos = torch.zeros_like(dos)
        if dg is not None:
            ddos[:B].copy_(dg)
        if ds is not None:
            ddos[B:].copy_(ds)
        Fn.dostransformer_bwd(P, G, self._cfg, m, ctx, ddos, None if dx is None else dx.float().contiguous(), sink)

    def forward(self, g):
        self._check_train_flags()
        return self._run(g)


class GraphnetworkBase(FusedModel):
    _cfg: Fn.ModelCfg
    _returns_x: bool

    def _extra_dead(self, g) -> Tuple[str, ...]:
        # Encoder picks node_encoder or node_encoder_prompt by input width
        # (graphnetwork_phonon.py:150-153, graphnetwork.py:96-99); the other one is dead for this run.
        expected = 118 if self._cfg.kind == "phonon" else 200
        width = g.x.shape[1] if g is not None else expected
        unused = "GN_encoder.node_encoder_prompt" if width == expected else "GN_encoder.node_encoder"
        return tuple(f"{unused}.{s}" for s in ("0.weight", "0.bias", "1.weight", "2.weight", "2.bias"))

    def _program_fwd(self, P, g, m):
        dos, xL, ctx = Fn.graphnetwork_fwd(P, self._cfg, g, m)
        return (dos, xL, ctx) if self._returns_x else (dos, ctx)

    def _program_bwd(self, P, G, m, saved, grads, sink):
        ddos = grads[0]
        dx = grads[1] if self._returns_x else None
        if ddos is None:
            ddos = torch.zeros(m.num_graphs, self._cfg.S, device=P["embeddings.weight"].device)
        Fn.graphnetwork_bwd(P, G, self._cfg, m, saved, ddos.float().contiguous(),
                            None if dx is None else dx.float().contiguous(), sink)

    def forward(self, g):
        out = self._run(g)
        return out if self._returns_x else out[0]
