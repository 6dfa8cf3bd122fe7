"""NFM — drop-in for rec_pangu/models/ranking/nfm.py:13-76.

logit = LR_Layer(data) + MLP(Bi_interaction_pooling(emb)); the MLP input is [B, D] (the dense columns only
enter through the LR term).  HIP forward: gather -> rp_fm_pool_fwd (Bi-interaction, one read of [B,F,D]) ->
MFMA linears -> loss kernel.
"""
from typing import Dict, List

from ..base_model import BaseModel, build_loss
from ..layers import LR_Layer, MLP, InnerProductLayer
from ..utils import get_dnn_input_dim


class NFM(BaseModel):
    def __init__(self, embedding_dim: int = 32, hidden_units: List[int] = [64, 64, 64],
                 loss_fun: str = 'torch.nn.BCELoss()', enc_dict: Dict[str, dict] = None):
        super(NFM, self).__init__(enc_dict, embedding_dim)
        self.hidden_units = hidden_units
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.lr = LR_Layer(enc_dict=self.enc_dict)
        self.inner_product_layer = InnerProductLayer(output="Bi_interaction_pooling")
        self.dnn_input_dim = get_dnn_input_dim(self.enc_dict, self.embedding_dim)
        self.dnn = MLP(input_dim=self.embedding_dim, output_dim=1, hidden_units=self.hidden_units,
                       hidden_activations='relu', dropout_rates=0)
        self.reset_parameters()

    def forward(self, data, is_training: bool = True):
        lr_logit = self.lr(data)
        bi = self.inner_product_layer(self.embedding_layer(data))  # [B, D]
        return self._finish([lr_logit, self.dnn(bi.view(lr_logit.shape[0], -1))], data, is_training, self.loss_fun)
