"""MLP — drop-in for rec_pangu/models/layers/deep.py:11-84.

The module tree is the reference's (`net` = nn.Sequential of Linear/BatchNorm1d/activation/Dropout
in the same order, so state_dict keys `net.<i>.{weight,bias}` and the init RNG draws are identical).
On a HIP device forward() walks `net` and runs every Linear (+ a directly following ReLU) as one
matrix-core launch with fused bias/ReLU epilogue (rp_linear_fwd); its backward is rp_linear_fwd on the
transposed weight + rp_linear_wgrad.  Dropout runs on rp_dropout_* (training mode; identity otherwise),
BatchNorm1d on rp_batchnorm_*; Tanh / Sigmoid / LeakyReLU(0.01) are epilogues of the same launch (round 5), other
activation modules are applied as they are (counted: hip.note_torch_path).
"""
import os
from typing import List, Union

import torch
import torch.nn as nn

from ... import functional as Fh
from .activation import Dice, get_activation


class MLP(nn.Module):
    def __init__(self, input_dim: int, output_dim: Union[int, None] = None, hidden_units: List[int] = [],
                 hidden_activations: Union[str, List[str]] = "ReLU", output_activation: Union[str, None] = None,
                 dropout_rates: Union[float, List[float]] = 0.1, batch_norm: bool = False, use_bias: bool = True):
        super(MLP, self).__init__()
        if output_dim is not None:
            assert isinstance(output_dim, int) and output_dim > 0, "output_dim must be an integer"
        assert isinstance(input_dim, int) and input_dim > 0, "input_dim must be an integer"
        assert isinstance(hidden_units, list) and all(isinstance(i, int) for i in hidden_units) and len(
            hidden_units) >= 1, "hidden_units must be a list of integers and with at least one element"
        if isinstance(hidden_activations, str):
            hidden_activations = [hidden_activations] * len(hidden_units)
        elif isinstance(hidden_activations, list):
            assert len(hidden_activations) == len(hidden_units), \
                "hidden_activations must have one element per hidden unit"
        else:
            raise TypeError("hidden_activations must be a string or a list of strings")
        if not isinstance(dropout_rates, list):
            dropout_rates = [dropout_rates] * len(hidden_units)
        else:
            assert len(dropout_rates) == len(hidden_units), "dropout_rates must have one element per hidden unit"

        dims = [input_dim] + hidden_units
        chain = []
        for i in range(len(dims) - 1):
            chain.append(nn.Linear(dims[i], dims[i + 1], bias=use_bias))
            if batch_norm:
                chain.append(nn.BatchNorm1d(dims[i + 1]))
            if hidden_activations[i]:
                chain.append(get_activation(hidden_activations[i]))
            if dropout_rates[i] > 0:
                chain.append(nn.Dropout(p=dropout_rates[i]))
        if output_dim is not None:
            chain.append(nn.Linear(dims[-1], output_dim, bias=use_bias))
        if output_activation is not None:
            chain.append(get_activation(output_activation))
        self.net = nn.Sequential(*chain)

    @staticmethod
    def _tail64_start(mods, x):
        """index of the first module of a trailing  [Linear(64,64), ReLU] x (1..3), Linear(64,1)  chain that directly
        follows a Linear(., 64) + ReLU (the reference's default hidden_units [64, 64, 64] with output_dim 1 and no
        dropout / BatchNorm: DeepFM, WDL, NFM), or -1."""
        from ... import hip
        n = len(mods)
        if n < 5 or not isinstance(mods[-1], nn.Linear) or mods[-1].out_features != 1 or mods[-1].in_features != 64:
            return -1
        j = n - 1
        layers = 0
        while j - 2 >= 0 and isinstance(mods[j - 1], nn.ReLU) and isinstance(mods[j - 2], nn.Linear) \
                and mods[j - 2].in_features == 64 and mods[j - 2].out_features == 64 and layers < 3:
            j -= 2
            layers += 1
        if layers == 0 or j < 2 or not (isinstance(mods[j - 1], nn.ReLU) and isinstance(mods[j - 2], nn.Linear)
                                        and mods[j - 2].out_features == 64):
            return -1
        if x.dim() != 2 or hip.get_matmul_precision() == "fp32":
            return -1
        return j

    def tail_bce(self, x, start: int, pending, addends, label, p_eps: float = 0.0, weight: float = 1.0):
        """(pred, loss) of BCE(sigmoid(sum(addends) + rest of the MLP from module `start` on), label) when that rest is exactly
        the fused tail ([Linear 64x64 + ReLU] x 1..3 -> Linear 64 -> 1 behind a Linear+ReLU whose ReluLink is `pending`) and
        there are at most 3 other addends: the loss head then rides inside the tail's two launches (Fh.mlp_tail64_bce).
        None otherwise — the caller runs the MLP and the loss separately."""
        if not x.is_cuda or pending is None or len(addends) > 3 or os.environ.get("RP_TAIL_BCE", "1") == "0":
            return None
        mods = list(self.net)
        if start <= 0 or self._tail64_start(mods, x) != start:
            return None
        if any(a.numel() != x.shape[0] or a.dtype is not torch.float32 for a in addends) or label.numel() != x.shape[0]:
            return None
        hidden = [(mods[j].weight, mods[j].bias) for j in range(start, len(mods) - 1, 2)]
        return Fh.mlp_tail64_bce(x, pending, hidden, (mods[-1].weight, mods[-1].bias), addends, label, p_eps, weight)

    def first_linear_relu(self):
        """the first Linear when it is directly followed by a ReLU (what the fused gather + Linear launch replaces)"""
        mods = list(self.net)
        if len(mods) >= 2 and isinstance(mods[0], nn.Linear) and isinstance(mods[1], nn.ReLU):
            return mods[0]
        return None

    def forward(self, x, fm_link=None, start: int = 0, pending=None):
        """`fm_link` (DeepFM on HIP): lets the first Linear's dgrad absorb the FM part of the embedding gradient.
        `start` / `pending`: continue after the first `start` modules (x is then the output of a Linear+ReLU that ran
        elsewhere — the fused gather + Linear launch — and `pending` its ReluLink)."""
        if not x.is_cuda:
            return self.net(x)  # BASELINE config 0 (CPU plumbing)
        mods = list(self.net)
        i = start
        tail_at = self._tail64_start(mods, x)
        while i < len(mods):
            m = mods[i]
            if i == tail_at and pending is not None:
                # everything from here on is [Linear 64x64 + ReLU] x L -> Linear 64 -> 1: one launch each way
                hidden = [(mods[j].weight, mods[j].bias) for j in range(i, len(mods) - 1, 2)]
                return Fh.mlp_tail64(x, pending, hidden, (mods[-1].weight, mods[-1].bias))
            if isinstance(m, nn.Linear):
                # the activation module behind a Linear rides in the GEMM's epilogue: ReLU (with the mask hand-off to the
                # next layer's dgrad), Tanh, Sigmoid, LeakyReLU(0.01) — activation.py:37-59 hands these out by name
                code = Fh.act_code(mods[i + 1]) if i + 1 < len(mods) else None
                fuse_relu = code == Fh.ACT_RELU
                out_link = Fh.ReluLink() if fuse_relu else None
                x = Fh.linear_act(x, m.weight, m.bias, code if code is not None else Fh.ACT_NONE,
                                  fm_link=fm_link if i == 0 else None, in_link=pending, out_link=out_link)
                pending = out_link
                i += 2 if code is not None else 1
            elif isinstance(m, nn.BatchNorm1d) and x.dim() == 2:
                x = Fh.batch_norm(x, m)
                pending = None
                i += 1
            elif isinstance(m, nn.Dropout):
                if self.training and 0 < m.p < 1:
                    x = Fh.dropout(x, m.p)  # rp_dropout_*: Philox mask, saved for the backward
                    pending = None
                elif self.training and m.p >= 1:
                    x = m(x)
                    pending = None
                i += 1
            elif Fh.act_code(m) is not None and x.dim() == 2 and x.dtype is torch.float32:
                # an activation that does not sit right behind a Linear (behind a BatchNorm1d, say): its own launch
                x = Fh.activation(x, Fh.act_code(m))
                pending = None
                i += 1
            else:
                if not isinstance(m, (nn.Identity, Dice)):  # (Dice dispatches to its own kernels)
                    from ... import hip
                    hip.note_torch_path(f"{type(m).__name__} inside an MLP (epilogues / launches exist for ReLU, Tanh, Sigmoid, "
                                        "LeakyReLU(0.01); BatchNorm1d, Dropout and Dice have kernels of their own)")
                x = m(x)
                pending = None
                i += 1
        return x
