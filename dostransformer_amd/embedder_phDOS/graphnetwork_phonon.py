"""MI355X drop-in for ``Graphnetwork_phonon`` (`embedder_phDOS/graphnetwork_phonon.py:14-72`): the
message-passing stack + energy embedding + 2-layer MLP head, no transformer.  ``forward(g) -> dos``.
(`Graphnetwork2_phonon` crashes upstream — Encoder arity `:114` vs `:148` — and is not provided.)"""
from torch import nn

from .. import functional as Fn
from .._blocks import Decoder, EdgeModel, Encoder, NodeModel, Processor
from .._models import GraphnetworkBase


class Graphnetwork_phonon(GraphnetworkBase):
    _returns_x = False

    def __init__(self, layers, n_atom_feats, n_bond_feats, n_hidden, dim_out, device):
        super().__init__()
        self.embeddings = nn.Embedding(51, n_hidden)
        self.GN_encoder = Encoder(n_atom_feats, n_bond_feats, n_hidden, prompt_branch=True)
        self.stacked_processor = nn.ModuleList(
            [Processor(EdgeModel(n_hidden), NodeModel(n_hidden)) for _ in range(layers)])
        self.GN_decoder = Decoder(n_hidden, n_hidden)
        self.device = device
        self.out_layer = nn.Sequential(nn.Linear(n_hidden * 2, n_hidden), nn.LeakyReLU(), nn.Linear(n_hidden, 1))
        self._cfg = Fn.ModelCfg("phonon", layers, 0, n_hidden, n_atom_feats, n_bond_feats, 51, True, "")
