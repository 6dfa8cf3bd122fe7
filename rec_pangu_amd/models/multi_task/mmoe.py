"""MMOE — drop-in for rec_pangu/models/multi_task/mmoe.py:14-130.

Faithful to the reference, defects included: `experts` ~ U[0,1) is left untouched by
the xavier pass; the per-task gates are N(0,1) tensors that are NOT registered parameters — they are
not in state_dict(), not in parameters() and never trained.  Here they are non-persistent buffers
(same visibility, but they follow .to(device), which replaces the reference's set_device hack).
Towers are Linear -> BatchNorm1d -> Dropout with no activation in between, then Linear -> Sigmoid;
loss = sum_t (1/T) * BCE(p_t + 1e-6, y_t).
"""
import torch

from ... import functional as Fh
from ..base_model import BaseModel
from ..utils import get_feature_num, get_linear_input
from .towers import build_towers, run_towers, weighted_bce


class MMOE(BaseModel):
    def __init__(self, num_task=2, n_expert=3, embedding_dim=40, mmoe_hidden_dim=128, expert_activation=None,
                 hidden_dim=[128, 64], dropouts=[0.2, 0.2], enc_dict=None, device=None):
        super(MMOE, self).__init__(enc_dict, embedding_dim)
        self.enc_dict = enc_dict
        self.num_task = num_task
        self.n_expert = n_expert
        self.mmoe_hidden_dim = mmoe_hidden_dim
        self.expert_activation = expert_activation
        self.hidden_dim = hidden_dim
        self.dropouts = dropouts
        self.num_sparse_fea, self.num_dense_fea = get_feature_num(self.enc_dict)
        hidden_size = self.num_sparse_fea * self.embedding_dim + self.num_dense_fea

        # RNG draws in the reference's order: experts, experts_bias, (gate rand + normal_) x T, gate biases
        self.experts = torch.nn.Parameter(torch.rand(hidden_size, mmoe_hidden_dim, n_expert), requires_grad=True)
        self.experts_bias = torch.nn.Parameter(torch.rand(mmoe_hidden_dim, n_expert), requires_grad=True)
        gates = [torch.rand(hidden_size, n_expert) for _ in range(num_task)]
        for g in gates:
            g.normal_(0, 1)
        gates_bias = [torch.rand(n_expert) for _ in range(num_task)]
        for t in range(num_task):
            self.register_buffer(f"_gate_{t}", gates[t], persistent=False)
            self.register_buffer(f"_gate_bias_{t}", gates_bias[t], persistent=False)

        build_towers(self, num_task, mmoe_hidden_dim, hidden_dim, dropouts)
        self.set_device(device)
        self.apply(self._init_weights)

    @property
    def gates(self):
        return [getattr(self, f"_gate_{t}") for t in range(self.num_task)]

    @property
    def gates_bias(self):
        return [getattr(self, f"_gate_bias_{t}") for t in range(self.num_task)]

    def set_device(self, device):
        """Kept for API compatibility (mmoe.py:64-68): the gates are buffers here and move with .to()."""
        if device is not None:
            for t in range(self.num_task):
                setattr(self, f"_gate_{t}", getattr(self, f"_gate_{t}").to(device))
                setattr(self, f"_gate_bias_{t}", getattr(self, f"_gate_bias_{t}").to(device))

    def forward(self, data, is_training=True):
        if self.on_hip and self.expert_activation is None:
            return self._forward_hip(data, is_training)
        if self.on_hip:
            x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
            h = self.experts.shape[0]
            hidden = x[:, :h] if x.shape[1] != h else x
        else:
            hidden = torch.cat([self.embedding_layer(data).flatten(start_dim=1),
                                get_linear_input(self.enc_dict, data)], axis=-1)

        experts_out = torch.einsum('ij, jkl -> ikl', hidden, self.experts) + self.experts_bias
        if self.expert_activation is not None:
            experts_out = self.expert_activation(experts_out)
        outs = []
        for gate, gate_bias in zip(self.gates, self.gates_bias):
            gate_out = torch.softmax(hidden @ gate + gate_bias, dim=-1)
            outs.append((experts_out * gate_out.unsqueeze(1)).sum(dim=2))

        return run_towers(self, outs, data, is_training, p_eps=1e-6)

    def _forward_hip(self, data, is_training):
        """HIP path: gather+concat -> ONE fp32-MFMA GEMM over [experts | gates] -> gate softmax + combine kernel
        -> towers (Linear on the MFMA kernel, BatchNorm1d on rp_batchnorm_*; Dropout as it is) -> sigmoid+BCE(p+1e-6) kernel."""
        x, _ = self.embedding_layer.gather_concat(data, self._dense_list(data), want_fm=False)
        h, K, E, T = self.experts.shape[0], self.mmoe_hidden_dim, self.n_expert, self.num_task
        N = K * E + T * E
        pack = self.__dict__.get("_pack")
        if pack is None or pack.shape != (x.shape[1], N) or pack.device != x.device:
            # [experts | gates] in the input-major layout the reference keeps them in, rows beyond h zero: filled from the
            # parameters by library launches every step (Fh.mmoe_project); created once, outside any captured step
            pack = self.__dict__["_pack"] = torch.zeros((x.shape[1], N), dtype=torch.float32, device=x.device)
        if all(g.shape == (h, E) for g in self.gates) and self.experts_bias.numel() == K * E:
            z = Fh.mmoe_project(x, pack, self.experts, self.experts_bias, list(self.gates), list(self.gates_bias))
        else:
            w_cat = torch.cat([self.experts.reshape(h, K * E)] + self.gates, dim=1)
            b_cat = torch.cat([self.experts_bias.reshape(-1)] + self.gates_bias)
            z = Fh.linear_input_major(x, w_cat, b_cat)
        mix = Fh.mmoe_combine(z, K, E, T)  # [T, B, K]
        return run_towers(self, [mix[i] for i in range(T)], data, is_training, p_eps=1e-6)

    def loss(self, task_outputs, data, weight=None):
        return weighted_bce(task_outputs, data, self.num_task, p_eps=1e-6, weight=weight)
