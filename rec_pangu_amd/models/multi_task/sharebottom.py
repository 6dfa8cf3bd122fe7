"""ShareBottom — drop-in for rec_pangu/models/multi_task/sharebottom.py:14-98.

Shared embedding bottom, one tower per task fed with cat(flatten(emb), dense) directly.  As in the reference the
xavier pass runs BEFORE the towers exist (sharebottom.py:38), so only the embedding tables get it and the tower
Linears keep nn.Linear's default init; there is no `device` argument.  Loss: sum_t (1/T) BCE(p_t, y_t).

HIP path: one gather launch writes the shared [B, ldx] input once; each tower's first Linear reads it in place.
"""
from typing import Dict, List

import torch

from ..base_model import BaseModel
from ..utils import get_feature_num, get_linear_input
from .towers import build_towers, run_towers, weighted_bce


class ShareBottom(BaseModel):
    def __init__(self, num_task: int = 2, embedding_dim: int = 40, hidden_units: List[int] = [128, 64],
                 dropouts: List[float] = [0.2, 0.2], enc_dict: Dict[str, dict] = None):
        super(ShareBottom, self).__init__(enc_dict, embedding_dim)
        self.enc_dict = enc_dict
        self.num_task = num_task
        self.hidden_dim = hidden_units
        self.dropouts = dropouts
        self.num_sparse_fea, self.num_dense_fea = get_feature_num(self.enc_dict)
        hidden_size = self.num_sparse_fea * self.embedding_dim + self.num_dense_fea
        self.apply(self._init_weights)
        build_towers(self, num_task, hidden_size, hidden_units, dropouts)

    def forward(self, data, is_training=True):
        if self.on_hip:
            out, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
        else:
            out = torch.cat([self.embedding_layer(data).flatten(start_dim=1),
                             get_linear_input(self.enc_dict, data)], axis=-1)
        return run_towers(self, [out] * self.num_task, data, is_training)

    def loss(self, task_outputs, data, weight=None):
        return weighted_bce(task_outputs, data, self.num_task, weight=weight)
