"""AutoInt — drop-in for rec_pangu/models/ranking/autoint.py:14-88:
logit = fc(flatten(self-attention stack over the field embeddings)) + MLP + LR."""
from typing import Dict, List

import torch
from torch import nn

from ... import functional as Fh
from ..base_model import BaseModel, build_loss
from ..layers import MLP, LR_Layer, MultiHeadSelfAttention
from ..utils import get_feature_num, get_linear_input


class AutoInt(BaseModel):
    def __init__(self, embedding_dim: int = 32, dnn_hidden_units: List[int] = [64, 64, 64], attention_layers: int = 1,
                 num_heads: int = 1, attention_dim: int = 8, loss_fun: str = 'torch.nn.BCELoss()',
                 enc_dict: Dict[str, dict] = None):
        super(AutoInt, self).__init__(enc_dict, embedding_dim)
        self.dnn_hidden_units = dnn_hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.num_sparse, self.num_dense = get_feature_num(self.enc_dict)
        self.lr_layer = LR_Layer(enc_dict=enc_dict)
        self.dnn = MLP(input_dim=self.embedding_dim * self.num_sparse + self.num_dense, output_dim=1,
                       hidden_units=self.dnn_hidden_units)
        self.self_attention = nn.Sequential(
            *[MultiHeadSelfAttention(self.embedding_dim if i == 0 else num_heads * attention_dim,
                                     attention_dim=attention_dim, num_heads=num_heads, align_to="output")
              for i in range(attention_layers)])
        self.fc = nn.Linear(self.num_sparse * attention_dim * num_heads, 1)
        self.reset_parameters()

    def forward(self, data, is_training=True):
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            F, D = self.num_sparse, self.embedding_dim
            link = getattr(self.embedding_layer, "_fm_link", None)
            if self.dnn is not None and link is not None and x.requires_grad and torch.is_grad_enabled():
                # the fields as [B, F, D] tokens: a packed copy by a library launch; its gradient joins the MLP's dX inside the
                # gather's backward (Fh.token_view) instead of through autograd's zero-filled slice gradient and ATen sum
                tokens = Fh.token_view(x, F, D, link)
                mlp_link = link  # (round 6: the first Linear's dgrad inside the gather's backward, the token gradient its dx operand)
            else:
                tokens = x[:, :F * D].unflatten(1, (F, D))
                mlp_link = None
            att = self.self_attention(tokens).flatten(start_dim=1)
            logits = [Fh.linear_act(att, self.fc.weight, self.fc.bias, Fh.ACT_NONE)]
            if self.dnn is not None:
                logits.append(self.dnn(x, fm_link=mlp_link))
            if self.lr_layer is not None:
                logits.append(self.lr_layer(data))
            return self._finish(logits, data, is_training, self.loss_fun)
        feature_emb = self.embedding_layer(data)
        logits = [self.fc(self.self_attention(feature_emb).flatten(start_dim=1))]
        if self.dnn is not None:
            dense_input = get_linear_input(self.enc_dict, data)
            logits.append(self.dnn(torch.cat([feature_emb.flatten(start_dim=1), dense_input], dim=1)))
        if self.lr_layer is not None:
            logits.append(self.lr_layer(data))
        return self._finish(logits, data, is_training, self.loss_fun)
