"""LR_Layer — drop-in for rec_pangu/models/layers/shallow.py:14-27 (the first-order / "wide" term).

Its own EmbeddingLayer(dim=1) -> [B,F], concatenated with the dense columns -> Linear(F+ND, 1).
On a HIP device the dim-1 gather + concat is the same fused multi-table kernel (scalar row path,
one lane per row) writing the [B, F+ND] Linear input directly, and the Linear is the MFMA kernel.
"""
import torch
from torch import nn

from ... import functional as Fh
from ..utils import get_dnn_input_dim, get_linear_input, dense_columns
from .embedding import make_embedding_layer


class LR_Layer(nn.Module):
    def __init__(self, enc_dict):
        super(LR_Layer, self).__init__()
        self.enc_dict = enc_dict
        self.emb_layer = make_embedding_layer(self.enc_dict, 1)
        self.dnn_input_dim = get_dnn_input_dim(self.enc_dict, 1)
        self.fc = nn.Linear(self.dnn_input_dim, 1)

    def forward(self, data):
        if self.fc.weight.is_cuda:
            x, _ = self.emb_layer.gather_concat(data, [data[c] for c in dense_columns(self.enc_dict)], want_fm=False,
                                                pad_to=4)
            return Fh.linear_act(x, self.fc.weight, self.fc.bias, Fh.ACT_NONE)
        sparse_emb = self.emb_layer(data).squeeze(-1)
        dense_input = get_linear_input(self.enc_dict, data)
        return self.fc(torch.cat((sparse_emb, dense_input), dim=1))
