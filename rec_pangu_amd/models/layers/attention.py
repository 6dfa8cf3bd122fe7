"""Field self-attention of AutoInt — drop-in for rec_pangu/models/layers/attention.py:12-101.

Quirks of the reference that are kept on purpose: heads are split by a RAW
.view(B*H, -1, a) of the [B,T,H*a] projections (:73-75), there is no 1/sqrt(d) scale unless
use_scale, the residual uses W_res only when input_dim != H*a, and ReLU is always applied (:94).
"""
import numpy as np
import torch
from torch import nn


class ScaledDotProductAttention(nn.Module):
    def __init__(self, dropout_rate=0.):
        super(ScaledDotProductAttention, self).__init__()
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None
        self.softmax = nn.Softmax(dim=2)

    def forward(self, W_q, W_k, W_v, scale=None, mask=None):
        scores = torch.bmm(W_q, W_k.transpose(1, 2))
        if scale:
            scores = scores / scale
        if mask:
            scores = scores.masked_fill_(mask, -np.inf)
        attention = self.softmax(scores)
        if self.dropout is not None:
            attention = self.dropout(attention)
        return torch.bmm(attention, W_v), attention


class MultiHeadAttention(nn.Module):
    def __init__(self, input_dim, attention_dim=None, num_heads=1, dropout_rate=0., use_residual=True,
                 use_scale=False, layer_norm=False, align_to="input"):
        super(MultiHeadAttention, self).__init__()
        if attention_dim is None:
            attention_dim = input_dim // num_heads
        self.attention_dim = attention_dim
        self.output_dim = num_heads * attention_dim
        self.num_heads = num_heads
        self.use_residual = use_residual
        self.align_to = align_to
        self.scale = attention_dim ** 0.5 if use_scale else None
        self.W_q = nn.Linear(input_dim, self.output_dim, bias=False)
        self.W_k = nn.Linear(input_dim, self.output_dim, bias=False)
        self.W_v = nn.Linear(input_dim, self.output_dim, bias=False)
        self.W_res = None
        if input_dim != self.output_dim:
            if align_to == "output":
                self.W_res = nn.Linear(input_dim, self.output_dim, bias=False)
            elif align_to == "input":
                self.W_res = nn.Linear(self.output_dim, input_dim, bias=False)
        self.dot_product_attention = ScaledDotProductAttention(dropout_rate)
        self.layer_norm = nn.LayerNorm(self.output_dim) if layer_norm else None
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None

    def forward(self, query, key, value, mask=None):
        residual = query
        B = query.size(0)
        q = self.W_q(query).view(B * self.num_heads, -1, self.attention_dim)
        k = self.W_k(key).view(B * self.num_heads, -1, self.attention_dim)
        v = self.W_v(value).view(B * self.num_heads, -1, self.attention_dim)
        if mask:
            mask = mask.repeat(self.num_heads, 1, 1)
        output, attention = self.dot_product_attention(q, k, v, self.scale, mask)
        output = output.view(B, -1, self.output_dim)
        if self.W_res is not None:
            if self.align_to == "output":
                residual = self.W_res(residual)
            elif self.align_to == "input":
                output = self.W_res(output)
        if self.dropout is not None:
            output = self.dropout(output)
        if self.use_residual:
            output = output + residual
        if self.layer_norm is not None:
            output = self.layer_norm(output)
        return output.relu(), attention


class MultiHeadSelfAttention(MultiHeadAttention):
    def forward(self, X):
        output, _ = super(MultiHeadSelfAttention, self).forward(X, X, X)
        return output
