"""LR — counterpart of rec_pangu/models/ranking/lr.py:12-51: pred = sigmoid(LR_Layer(data)).

The reference class cannot be constructed: it subclasses `nn.Module` (lr.py:12), not BaseModel, and its constructor ends
with `self.reset_parameters()` (lr.py:28) — a method nn.Module does not have: AttributeError.  This one implements what that
file intends — no `embedding_layer`, a single `lr_layer`, BaseModel's initialisation — so BenchmarkTrainer's default ranking
list can run it.
"""
from typing import Dict

from torch import nn

from ..base_model import BaseModel, build_loss
from ..layers import LR_Layer


class LR(BaseModel):
    def __init__(self, loss_fun: str = 'torch.nn.BCELoss()', enc_dict: Dict[str, dict] = None):
        nn.Module.__init__(self)
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.lr_layer = LR_Layer(enc_dict=self.enc_dict)
        self.reset_parameters()

    @property
    def on_hip(self) -> bool:
        return self.lr_layer.fc.weight.is_cuda

    def forward(self, data, is_training: bool = True):
        return self._finish([self.lr_layer(data)], data, is_training, self.loss_fun)
