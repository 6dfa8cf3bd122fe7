"""DCN — drop-in for rec_pangu/models/ranking/dcn.py:14-68.

sigmoid(fc(CrossNet(cat(flatten(emb), dense)))): like the reference there is NO deep branch
(`hidden_units` is accepted and unused, dcn.py:17).
"""
from typing import Dict, List

import torch
from torch import nn

from ..base_model import BaseModel, build_loss
from ..layers import CrossNet
from ..utils import get_feature_num, get_linear_input


class DCN(BaseModel):
    def __init__(self, embedding_dim: int = 32, hidden_units: List[int] = [64, 64, 64], crossing_layers: int = 3,
                 loss_fun: str = 'torch.nn.BCELoss()', enc_dict: Dict[str, dict] = None):
        super(DCN, self).__init__(enc_dict, embedding_dim)
        self.dnn_hidden_units = hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.num_sparse, self.num_dense = get_feature_num(self.enc_dict)
        input_dim = self.num_sparse * self.embedding_dim + self.num_dense
        self.crossnet = CrossNet(input_dim, crossing_layers)
        self.fc = nn.Linear(input_dim, 1)
        self.reset_parameters()

    def forward(self, data, is_training=True):
        if self.on_hip:
            # 3 launches: gather+concat -> all cross layers + fc (X_L never leaves the chip) -> sigmoid+BCE
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            return self._finish([self.crossnet(x, fc=self.fc)], data, is_training, self.loss_fun)
        feature_emb = self.embedding_layer(data)
        x = torch.cat([feature_emb.flatten(start_dim=1), get_linear_input(self.enc_dict, data)], dim=1)
        return self._finish([self.fc(self.crossnet(x))], data, is_training, self.loss_fun)
