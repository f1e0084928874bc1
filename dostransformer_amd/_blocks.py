"""Parameter containers of the message-passing stack.  They reproduce the reference's module tree
(hence its ``state_dict`` keys and, with the same seed, its initial values) but hold no arithmetic:
the fused programs in ``functional.py`` read the parameters directly."""
from __future__ import annotations

from torch import nn


def mlp_prelu(n_in: int, n_hidden: int) -> nn.Sequential:
    """Linear -> PReLU -> Linear (the three encoders, `DOSTransformer_phonon.py:129-130`)."""
    return nn.Sequential(nn.Linear(n_in, n_hidden), nn.PReLU(), nn.Linear(n_hidden, n_hidden))


def mlp_ln(n_in: int, n_hidden: int) -> nn.Sequential:
    """Linear -> LayerNorm -> PReLU -> Linear (`DOSTransformer_phonon.py:193,203-204`)."""
    return nn.Sequential(nn.Linear(n_in, n_hidden * 2), nn.LayerNorm(n_hidden * 2), nn.PReLU(),
                         nn.Linear(n_hidden * 2, n_hidden))


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} only stores parameters; call the enclosing model "
                           f"(its forward is one fused libdosx program)")


class Encoder(_Holder):
    """`DOSTransformer_phonon.py:126-131`, `DOSTransformer.py:100-106`, `graphnetwork_phonon.py:132-141`,
    `graphnetwork.py:79-87`.  Creation order matters for same-seed initial values."""

    def __init__(self, n_atom_feats, n_bond_feats, n_hidden, n_global_feats=None, prompt_branch=False):
        super().__init__()
        self.node_encoder = mlp_prelu(n_atom_feats, n_hidden)
        if prompt_branch:
            self.node_encoder_prompt = mlp_prelu(n_atom_feats + n_hidden // 2, n_hidden)
        self.edge_encoder = mlp_prelu(n_bond_feats, n_hidden)
        if n_global_feats is not None:
            self.global_encoder = mlp_prelu(n_global_feats, n_hidden)


class EdgeModel(_Holder):
    def __init__(self, n_hidden):
        super().__init__()
        self.edge_mlp = mlp_ln(n_hidden * 3, n_hidden)


class NodeModel(_Holder):
    def __init__(self, n_hidden):
        super().__init__()
        self.node_mlp_1 = mlp_ln(n_hidden * 2, n_hidden)     # never used upstream either -> grad stays None
        self.node_mlp_2 = mlp_ln(n_hidden * 2, n_hidden)


class Processor(_Holder):
    def __init__(self, edge_model=None, node_model=None):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model


class Decoder(_Holder):
    def __init__(self, n_in, n_hidden):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(n_in, n_hidden))
