"""MLMMOE — drop-in for rec_pangu/models/multi_task/mlmmoe.py:13-146.

MMOE with a "level" mixing stage in front of the per-task gates:
    experts_out[b,k,e] = hidden . experts[:,k,e] + experts_bias[k,e]
    level_out[b,k,j]   = sum_e experts_out[b,k,e] * softmax(level_gates[j], dim=0)[e]       (E input-independent gates)
    out_t[b,k]         = sum_j level_out[b,k,j] * softmax(hidden . gates[t] + gates_bias[t])[j]
Faithful to the reference, defects included: `level_gates`, `gates` and `gates_bias` are python lists of tensors
that never reach parameters()/state_dict() and are therefore never trained (same defect as MMOE); here
they are non-persistent buffers so that .to(device) moves them.  experts ~ N(0,1).  Loss without the +1e-6.

HIP path (expert_activation=None): the level stage is input-independent and linear, so it is folded into the
expert weights in weight space — experts' = experts . Lm, bias' = bias . Lm with Lm[e,j] = softmax(level_gates[j])[e]
— and the rest is exactly the MMOE kernels: one MFMA GEMM over [experts' | gates] + the gate-softmax/combine kernel.
"""
import torch

from ... import functional as Fh
from ..base_model import BaseModel
from ..utils import get_feature_num, get_linear_input
from .towers import build_towers, run_towers, weighted_bce


class MLMMOE(BaseModel):
    def __init__(self, num_task=2, n_expert=3, embedding_dim=40, mmoe_hidden_dim=128, expert_activation=None,
                 hidden_dim=[128, 64], dropouts=[0.2, 0.2], enc_dict=None, device=None):
        super(MLMMOE, self).__init__(enc_dict, embedding_dim)
        self.enc_dict = enc_dict
        self.num_task = num_task
        self.n_expert = n_expert
        self.mmoe_hidden_dim = mmoe_hidden_dim
        self.expert_activation = expert_activation
        self.hidden_dim = hidden_dim
        self.dropouts = dropouts
        self.num_sparse_fea, self.num_dense_fea = get_feature_num(self.enc_dict)
        hidden_size = self.num_sparse_fea * self.embedding_dim + self.num_dense_fea

        # RNG draws in the reference's order
        self.experts = torch.nn.Parameter(torch.rand(hidden_size, mmoe_hidden_dim, n_expert), requires_grad=True)
        self.experts.data.normal_(0, 1)
        self.experts_bias = torch.nn.Parameter(torch.rand(mmoe_hidden_dim, n_expert), requires_grad=True)
        level_gates = [torch.rand(n_expert, 1) for _ in range(n_expert)]
        gates = [torch.rand(hidden_size, n_expert) for _ in range(num_task)]
        for g in gates:
            g.normal_(0, 1)
        gates_bias = [torch.rand(n_expert) for _ in range(num_task)]
        for j in range(n_expert):
            self.register_buffer(f"_level_gate_{j}", level_gates[j], persistent=False)
        for t in range(num_task):
            self.register_buffer(f"_gate_{t}", gates[t], persistent=False)
            self.register_buffer(f"_gate_bias_{t}", gates_bias[t], persistent=False)
        build_towers(self, num_task, mmoe_hidden_dim, hidden_dim, dropouts)
        self.set_device(device)
        self.apply(self._init_weights)

    level_gates = property(lambda self: [getattr(self, f"_level_gate_{j}") for j in range(self.n_expert)])
    gates = property(lambda self: [getattr(self, f"_gate_{t}") for t in range(self.num_task)])
    gates_bias = property(lambda self: [getattr(self, f"_gate_bias_{t}") for t in range(self.num_task)])

    def set_device(self, device):
        """Kept for API compatibility (mlmmoe.py:66-72): the gates are buffers here and move with .to()."""
        if device is not None:
            for name, buf in list(self.named_buffers(recurse=False)):
                setattr(self, name, buf.to(device))

    def level_matrix(self):
        """Lm [E, E]: column j = softmax(level_gates[j], dim=0)."""
        return torch.cat([torch.softmax(g, dim=0) for g in self.level_gates], dim=1)

    def forward(self, data, is_training=True):
        h, K, E = self.experts.shape
        T = self.num_task
        Lm = self.level_matrix()
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            if self.expert_activation is None:
                w_lv = (self.experts.reshape(h * K, E) @ Lm).view(h, K * E)
                b_lv = (self.experts_bias @ Lm).reshape(-1)
                w_cat = torch.cat([w_lv] + self.gates, dim=1)
                b_cat = torch.cat([b_lv] + self.gates_bias)
                mix = Fh.mmoe_combine(Fh.linear_input_major(x, w_cat, b_cat), K, E, T)  # [T, B, K]
                return run_towers(self, [mix[i] for i in range(T)], data, is_training)
            hidden = x[:, :h] if x.shape[1] != h else x
        else:
            hidden = torch.cat([self.embedding_layer(data).flatten(start_dim=1),
                                get_linear_input(self.enc_dict, data)], axis=-1)
        experts_out = torch.einsum('ij, jkl -> ikl', hidden, self.experts) + self.experts_bias
        if self.expert_activation is not None:
            experts_out = self.expert_activation(experts_out)
        # same op sequence as mlmmoe.py:94-99 (one einsum per level gate, then cat): the towers' pre-BatchNorm bias
        # gradients are pure rounding noise that Adam amplifies, so the CPU plumbing path keeps the rounding too
        level_out = torch.cat([torch.einsum('abc, cd -> abd', experts_out, Lm[:, j:j + 1]) for j in range(E)],
                              axis=-1)
        outs = []
        for gate, gate_bias in zip(self.gates, self.gates_bias):
            gate_out = torch.softmax(hidden @ gate + gate_bias, dim=-1)
            outs.append((level_out * gate_out.unsqueeze(1)).sum(dim=2))
        return run_towers(self, outs, data, is_training)

    def loss(self, task_outputs, data, weight=None):
        return weighted_bce(task_outputs, data, self.num_task, weight=weight)
