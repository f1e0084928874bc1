"""MI355X drop-in for ``embedder_eDOS/DOSTransformer.py`` (reference `:12-93`): 201 energy bins,
precomputed ``edge_attr``, a ``glob`` encoder feeding the decoder, ``scatter_sum`` aggregation
(`:187`) and the upstream attribute spelling ``promt_token`` (`:20`)."""
from torch import nn

from .. import functional as Fn
from .._blocks import Decoder, EdgeModel, Encoder, NodeModel, Processor
from .._models import DOSTransformerBase
from ..layers import TransformerEncoder


class DOSTransformer(DOSTransformerBase):
    def __init__(self, layers, t_layers, n_atom_feats, n_bond_feats, n_glob_feats, n_hidden, device, attn_drop):
        super().__init__()
        self.embeddings = nn.Embedding(201, n_hidden)
        self.promt_token = nn.Embedding(7, n_hidden // 2)
        self.GN_encoder = Encoder(n_atom_feats, n_bond_feats, n_hidden, n_global_feats=n_glob_feats)
        self.stacked_processor = nn.ModuleList(
            [Processor(EdgeModel(n_hidden), NodeModel(n_hidden, aggr="sum")) for _ in range(layers)])
        for name in ("transformer", "transformer_self", "transformer_source"):
            setattr(self, name, TransformerEncoder(embed_dim=n_hidden, num_heads=1, layers=t_layers,
                                                   attn_dropout=attn_drop))
        self.GN_decoder = Decoder(n_hidden * 2, n_hidden)
        self.out_layer = nn.Linear(n_hidden, 1)
        self.fc_prompt = nn.Linear(n_hidden * 2 + n_hidden // 2, n_hidden)
        self.fc = nn.Linear(n_hidden * 2, n_hidden)
        self.device = device
        self._attn_drop = attn_drop
        if n_glob_feats != 2:
            raise ValueError("the reference reshapes glob to (-1, 2) (DOSTransformer.py:119): n_glob_feats must be 2")
        self._cfg = Fn.ModelCfg("edos", layers, t_layers, n_hidden, n_atom_feats, n_bond_feats, 201, False,
                                "promt_token.weight")
