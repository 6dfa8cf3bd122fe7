"""Activation lookup (reference: rec_pangu/models/layers/activation.py:37-59)."""
from torch import nn


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
