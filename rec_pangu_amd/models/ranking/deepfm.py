"""DeepFM — drop-in for rec_pangu/models/ranking/deepfm.py:13-67.

logit = FM second order + MLP(cat(flatten(emb), dense)); there is NO first-order term in the
reference and none here.  HIP forward = 1 gather launch (embeddings + dense columns + FM) ->
MLP GEMMs -> 1 loss launch; the [B,F,D] tensor, the flatten and the cat never exist.
"""
from typing import Dict, List

import torch

from ... import functional as Fh
from ..base_model import BaseModel, _is_plain_bce as is_plain_bce, build_loss
from ..layers import FM_Layer, MLP
from ..utils import get_dnn_input_dim, get_linear_input


class DeepFM(BaseModel):
    def __init__(self, embedding_dim: int = 32, hidden_units: List[int] = [64, 64, 64],
                 loss_fun: str = 'torch.nn.BCELoss()', enc_dict: Dict[str, dict] = None):
        super(DeepFM, self).__init__(enc_dict, embedding_dim)
        self.hidden_units = hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.fm = FM_Layer()
        self.dnn_input_dim = get_dnn_input_dim(self.enc_dict, self.embedding_dim)
        self.dnn = MLP(input_dim=self.dnn_input_dim, output_dim=1, hidden_units=self.hidden_units,
                       hidden_activations='relu', dropout_rates=0)
        self.reset_parameters()

    def forward(self, data, is_training=True):
        if self.on_hip:
            dense = self._dense_list(data)
            first = self.dnn.first_linear_relu()
            if first is not None and self.embedding_layer.gather_linear_fits(len(dense), first):
                # lookup + dense concat + FM second order + dnn.net.0 (+ ReLU) in ONE launch; the rest of the MLP follows
                link = Fh.ReluLink()
                h1, fm_out = self.embedding_layer.gather_linear(data, dense, first, link)
                if is_training and is_plain_bce(self.loss_fun):
                    # sigmoid + BCE (deepfm.py:61-66) inside the fused MLP tail's launches: pred and the gradients are
                    # bit-identical to the separate launches, the loss is the same sum in another order
                    out = self.dnn.tail_bce(h1, 2, link, [fm_out], data["label"].float())
                    if out is not None:
                        return {"pred": out[0], "loss": out[1]}
                return self._finish([fm_out, self.dnn(h1, start=2, pending=link)], data, is_training, self.loss_fun)
            x, fm_out = self.embedding_layer.gather_concat(data, dense, want_fm=True)
            link = getattr(self.embedding_layer, "_fm_link", None)
            if link is None or not fm_out.requires_grad:
                return self._finish([fm_out, self.dnn(x)], data, is_training, self.loss_fun)
            deep = self.dnn(x, fm_link=link)
            # tapped AFTER the MLP forward: its backward (d loss / d fm -> link) then runs before the MLP's
            return self._finish([Fh.tap_fm_grad(fm_out, link), deep], data, is_training, self.loss_fun)
        sparse_embedding = self.embedding_layer(data)
        dnn_input = torch.cat((sparse_embedding.flatten(start_dim=1), get_linear_input(self.enc_dict, data)), dim=1)
        return self._finish([self.fm(sparse_embedding), self.dnn(dnn_input)], data, is_training, self.loss_fun)
