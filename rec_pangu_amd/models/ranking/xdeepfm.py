"""xDeepFM — drop-in for rec_pangu/models/ranking/xdeepfm.py:13-79: logit = LR + CIN + MLP."""
from typing import Dict, List

import torch

from ..base_model import BaseModel, build_loss
from ..layers import MLP, LR_Layer, CompressedInteractionNet
from ..utils import get_feature_num, get_linear_input


class xDeepFM(BaseModel):
    def __init__(self, embedding_dim: int = 32, dnn_hidden_units: List[int] = [64, 64, 64],
                 cin_layer_units: List[int] = [16, 16, 16], loss_fun: str = 'torch.nn.BCELoss()',
                 enc_dict: Dict[str, dict] = None) -> None:
        super(xDeepFM, self).__init__(enc_dict, embedding_dim)
        self.embedding_dim = embedding_dim
        self.dnn_hidden_units = dnn_hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.num_sparse, self.num_dense = get_feature_num(self.enc_dict)
        self.dnn = MLP(input_dim=self.num_sparse * self.embedding_dim + self.num_dense, output_dim=1,
                       hidden_units=self.dnn_hidden_units)
        self.lr_layer = LR_Layer(enc_dict=self.enc_dict)
        self.cin = CompressedInteractionNet(self.num_sparse, cin_layer_units, output_dim=1)
        self.reset_parameters()

    def forward(self, data, is_training: bool = True):
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            F, D = self.num_sparse, self.embedding_dim
            link = getattr(self.embedding_layer, "_fm_link", None)
            if self.dnn is not None and link is not None and x.requires_grad and torch.is_grad_enabled():
                # the embedding block of the MLP input buffer as the CIN's [B, F D] rows (no copy: the kernels take the row
                # stride); its gradient joins the MLP's dX inside the gather's backward (Fh.token_alias) instead of through
                # autograd's zero-filled slice gradient and an ATen sum of two 450 MB tensors
                from ... import functional as Fh
                feature_emb = Fh.token_alias(x, F * D, link)
            else:
                feature_emb = x[:, :F * D].unflatten(1, (F, D))  # strided [B,F,D] view of the MLP input buffer
            logits = [self.lr_layer(data)] + list(self.cin(feature_emb, as_list=True))
            if self.dnn is not None:
                # (fm_link: the first Linear leaves its dgrad to the gather's backward where it fits — rp_embed_grad_gemm then
                #  takes the CIN's gradient of the embedding block as its dx operand: dX is never materialised)
                logits.append(self.dnn(x, fm_link=link if (link is not None and feature_emb.dim() == 2) else None))
            return self._finish(logits, data, is_training, self.loss_fun)
        feature_emb = self.embedding_layer(data)
        logits = [self.lr_layer(data), self.cin(feature_emb)]
        if self.dnn is not None:
            dense_input = get_linear_input(self.enc_dict, data)
            logits.append(self.dnn(torch.cat([feature_emb.flatten(start_dim=1), dense_input], dim=1)))
        return self._finish(logits, data, is_training, self.loss_fun)
