"""Dice + activation lookup — drop-in for rec_pangu/models/layers/activation.py:10-59."""
import torch
from torch import nn


class Dice(nn.Module):
    """activation.py:10-34:  p = sigmoid(BatchNorm1d(x, affine=False, eps, momentum=0.01)),  out = p x + (1 - p) alpha x.
    Same submodule / parameter names (`bn`, `alpha`) and registration order as the reference.  HIP-resident 2-D inputs
    run on rp_batchnorm_* + rp_dice_gate_* (rec_pangu_amd/functional.py: dice); a [B, L, N] input is normalised over
    the channel dimension 1 by nn.BatchNorm1d in the reference — that layout and CPU tensors take the torch ops."""

    def __init__(self, input_dim: int, eps: float = 1e-9):
        super(Dice, self).__init__()
        self.bn = nn.BatchNorm1d(input_dim, affine=False, eps=eps, momentum=0.01)
        self.alpha = nn.Parameter(torch.zeros(input_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.dim() == 2 and x.dtype is torch.float32:
            from ... import functional as Fh
            return Fh.dice(x, self.bn, self.alpha)
        if x.is_cuda:
            from ... import hip
            hip.note_torch_path("Dice on a non-2-D / non-fp32 HIP tensor")
        p = torch.sigmoid(self.bn(x))
        return p * x + (1 - p) * self.alpha * x


def get_activation(activation):
    if isinstance(activation, str):
        low = activation.lower()
        if low == "relu":
            return nn.ReLU()
        if low == "sigmoid":
            return nn.Sigmoid()
        if low == "tanh":
            return nn.Tanh()
        return getattr(nn, activation)()
    return activation
