"""OMOE — drop-in for rec_pangu/models/multi_task/omoe.py:13-112.

One input-independent gate shared by every task: gate = softmax(self.gate [E,1], dim=0);
gate_out[b,k] = sum_e (hidden . experts[:,k,e] + experts_bias[k,e]) * gate[e]; every tower gets the same
gate_out.  Init as the reference: experts ~ N(0,1), experts_bias/gate ~ U[0,1), then the xavier pass over
Embedding/Linear.  Loss: sum_t (1/T) BCE(p_t, y_t) (no +1e-6 here, unlike MMOE).

HIP path: the gate does not depend on the sample and there is no activation between the expert GEMM and the
mix (expert_activation=None, the default), so the mix is folded into the weights — W_eff[h,K] = experts . gate,
b_eff = experts_bias . gate, an [h,K,E]x[E] contraction in weight space — and ONE fp32-MFMA GEMM of 1/E the
reference's flops produces gate_out.  Gradients reach experts/experts_bias/gate through that contraction.
"""
import torch

from ... import functional as Fh
from ..base_model import BaseModel
from ..utils import get_feature_num, get_linear_input
from .towers import build_towers, run_towers, weighted_bce


class OMOE(BaseModel):
    def __init__(self, num_task=2, n_expert=3, embedding_dim=40, omoe_hidden_dim=128, expert_activation=None,
                 hidden_dim=[128, 64], dropouts=[0.2, 0.2], enc_dict=None, device=None):
        super(OMOE, self).__init__(enc_dict, embedding_dim)
        self.enc_dict = enc_dict
        self.num_task = num_task
        self.n_expert = n_expert
        self.omoe_hidden_dim = omoe_hidden_dim
        self.expert_activation = expert_activation
        self.hidden_dim = hidden_dim
        self.dropouts = dropouts
        self.num_sparse_fea, self.num_dense_fea = get_feature_num(self.enc_dict)
        hidden_size = self.num_sparse_fea * self.embedding_dim + self.num_dense_fea

        self.experts = torch.nn.Parameter(torch.rand(hidden_size, omoe_hidden_dim, n_expert), requires_grad=True)
        self.experts.data.normal_(0, 1)
        self.experts_bias = torch.nn.Parameter(torch.rand(omoe_hidden_dim, n_expert), requires_grad=True)
        self.gate = torch.nn.Parameter(torch.rand(n_expert, 1), requires_grad=True)
        build_towers(self, num_task, omoe_hidden_dim, hidden_dim, dropouts)
        self.apply(self._init_weights)

    def forward(self, data, is_training=True):
        gate = torch.softmax(self.gate, dim=0)  # [E,1]
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            h, K, E = self.experts.shape
            if self.expert_activation is None:
                w_eff = (self.experts.reshape(h * K, E) @ gate).view(h, K)
                b_eff = (self.experts_bias @ gate).view(K)
                gate_out = Fh.linear_input_major(x, w_eff, b_eff)
            else:
                z = Fh.linear_input_major(x, self.experts.reshape(h, K * E), self.experts_bias.reshape(-1))
                gate_out = (self.expert_activation(z.view(-1, K, E)) @ gate).squeeze(-1)
        else:
            hidden = torch.cat([self.embedding_layer(data).flatten(start_dim=1),
                                get_linear_input(self.enc_dict, data)], axis=-1)
            experts_out = torch.einsum('ij, jkl -> ikl', hidden, self.experts) + self.experts_bias
            if self.expert_activation is not None:
                experts_out = self.expert_activation(experts_out)
            gate_out = torch.einsum('abc, cd -> abd', experts_out, gate).squeeze(-1)
        return run_towers(self, [gate_out] * self.num_task, data, is_training)

    def loss(self, task_outputs, data, weight=None):
        return weighted_bce(task_outputs, data, self.num_task, weight=weight)
