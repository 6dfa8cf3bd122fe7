"""FM — drop-in for rec_pangu/models/ranking/fm.py:12-60: logit = FM second order only."""
from typing import Dict

from ..base_model import BaseModel, build_loss
from ..layers import FM_Layer


class FM(BaseModel):
    def __init__(self, embedding_dim: int = 32, loss_fun: str = 'torch.nn.BCELoss()',
                 enc_dict: Dict[str, dict] = None):
        super(FM, self).__init__(enc_dict, embedding_dim)
        self.loss_fun = build_loss(loss_fun)
        self.enc_dict = enc_dict
        self.fm = FM_Layer()
        self.reset_parameters()

    def forward(self, data, is_training: bool = True):
        if self.on_hip:
            _, fm_out = self.embedding_layer.gather_concat(data, [], want_fm=True, pad_to=4)
            return self._finish([fm_out], data, is_training, self.loss_fun)
        return self._finish([self.fm(self.embedding_layer(data))], data, is_training, self.loss_fun)
