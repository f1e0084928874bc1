"""MI355X drop-in for ``embedder_phDOS/DOSTransformer_phonon.py`` (reference `:14-119`).

Same constructor signature, parameter names/shapes (and, for a given seed, initial values: modules
are created in the upstream order), same ``forward(g) -> (dos_global, x, dos_system)``.  The forward
and backward are libdosx programs (see ``dostransformer_amd/functional.py``); aggregation is
``scatter_mean`` (`:209`), 51 energy bins (`:19`), 7 crystal-system prompts (`:21`).
"""
import torch
from torch import nn

from .. import functional as Fn
from .._blocks import Decoder, EdgeModel, Encoder, NodeModel, Processor
from .._models import DOSTransformerBase
from ..layers import TransformerEncoder


class DOSTransformer_phonon(DOSTransformerBase):
    def __init__(self, layers, t_layers, n_atom_feats, n_bond_feats, n_hidden, device, attn_drop):
        super().__init__()
        self.embeddings = nn.Embedding(51, n_hidden)
        self.prompt_token = nn.Embedding(7, n_hidden // 2)
        self.GN_encoder = Encoder(n_atom_feats, n_bond_feats, n_hidden)
        self.stacked_processor = nn.ModuleList(
            [Processor(EdgeModel(n_hidden), NodeModel(n_hidden)) for _ in range(layers)])
        for name in ("transformer", "transformer_self", "transformer_source"):
            setattr(self, name, TransformerEncoder(embed_dim=n_hidden, num_heads=1, layers=t_layers,
                                                   attn_dropout=attn_drop))
        self.GN_decoder = Decoder(n_hidden, n_hidden)
        self.alpha = nn.Parameter(torch.rand(1))          # unused upstream as well (`:40`)
        self.out_layer = nn.Linear(n_hidden, 1)
        self.fc = nn.Linear(n_hidden * 2, n_hidden)
        self.fc_prompt = nn.Linear(n_hidden * 2 + n_hidden // 2, n_hidden)
        self.device = device
        self._attn_drop = attn_drop
        self._cfg = Fn.ModelCfg("phonon", layers, t_layers, n_hidden, n_atom_feats, n_bond_feats, 51, True,
                                "prompt_token.weight")
