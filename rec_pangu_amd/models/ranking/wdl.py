"""WDL (Wide & Deep) — drop-in for rec_pangu/models/ranking/wdl.py:13-73.

logit = LR_Layer(data) ("wide": dim-1 tables + dense -> Linear) + MLP(cat(flatten(emb), dense)) ("deep").
HIP forward = 2 gather launches (dim-1 wide tables, dim-D deep tables; both write their consumer's input
directly) -> MFMA linears -> 1 loss launch.
"""
from typing import Dict, List

import torch

from ..base_model import BaseModel, build_loss
from ..layers import LR_Layer, MLP
from ..utils import get_dnn_input_dim, get_linear_input


class WDL(BaseModel):
    def __init__(self, embedding_dim: int = 32, hidden_units: List[int] = [64, 64, 64],
                 loss_fun: str = 'torch.nn.BCELoss()', enc_dict: Dict[str, dict] = None) -> None:
        super(WDL, self).__init__(enc_dict, embedding_dim)
        self.hidden_units = hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.lr = LR_Layer(enc_dict=self.enc_dict)
        self.dnn_input_dim = get_dnn_input_dim(self.enc_dict, self.embedding_dim)
        self.dnn = MLP(input_dim=self.dnn_input_dim, output_dim=1, hidden_units=self.hidden_units,
                       hidden_activations='relu', dropout_rates=0)
        self.reset_parameters()

    def forward(self, data, is_training: bool = True):
        wide_logit = self.lr(data)
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
        else:
            x = torch.cat([self.embedding_layer(data).flatten(start_dim=1), get_linear_input(self.enc_dict, data)],
                          dim=1)
        return self._finish([wide_logit, self.dnn(x)], data, is_training, self.loss_fun)
