"""MI355X drop-in for ``Graphnetwork`` (`embedder_eDOS/graphnetwork.py:10-43`): ``forward(g) -> (dos, x)``.
(`Graphnetwork2` crashes upstream — Encoder arity `:64` vs `:94` — and is not provided.)"""
from torch import nn

from .. import functional as Fn
from .._blocks import Decoder, EdgeModel, Encoder, NodeModel, Processor
from .._models import GraphnetworkBase


class Graphnetwork(GraphnetworkBase):
    _returns_x = True

    def __init__(self, layers, n_atom_feats, n_bond_feats, n_glob_feats, n_hidden, dim_out, device):
        super().__init__()
        self.embeddings = nn.Embedding(201, n_hidden)
        self.GN_encoder = Encoder(n_atom_feats, n_bond_feats, n_hidden, n_global_feats=n_glob_feats,
                                  prompt_branch=True)
        self.stacked_processor = nn.ModuleList(
            [Processor(EdgeModel(n_hidden), NodeModel(n_hidden, aggr="sum")) for _ in range(layers)])
        self.GN_decoder = Decoder(n_hidden * 2, n_hidden)
        self.device = device
        self.out_layer = nn.Sequential(nn.Linear(n_hidden * 2, n_hidden), nn.LeakyReLU(), nn.Linear(n_hidden, 1))
        self._cfg = Fn.ModelCfg("edos", layers, 0, n_hidden, n_atom_feats, n_bond_feats, 201, False, "")
