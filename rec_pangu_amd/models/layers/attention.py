"""Field self-attention of AutoInt — drop-in for rec_pangu/models/layers/attention.py:12-101.

Same classes, constructor arguments, sub-module names (W_q, W_k, W_v, W_res, dot_product_attention,
layer_norm, dropout -> same state_dict keys) and the reference's quirks, kept on purpose:
heads are split by a RAW .view(B*H, -1, a) of the [B,T,H*a] projections (attention.py:73-75), there is no
1/sqrt(d) scale unless use_scale, the residual goes through W_res only when input_dim != H*a, and ReLU is
always applied last (:94).

On a HIP device the AutoInt configuration (self-attention, align_to="output", no mask / dropout / LayerNorm,
residual on) runs as ONE launch per layer (rp_field_attention_fwd): projections, raw-view head split, scores,
softmax, PV, residual and ReLU for a sample all stay in LDS.  Any other configuration of the general
MultiHeadAttention module is composed from device ops as written in `_compose`.
"""
import numpy as np
import torch
from torch import nn

from ... import functional as Fh


class ScaledDotProductAttention(nn.Module):
    def __init__(self, dropout_rate=0.):
        super(ScaledDotProductAttention, self).__init__()
        self.dropout = nn.Dropout(dropout_rate) if dropout_rate > 0 else None
        self.softmax = nn.Softmax(dim=2)

    def forward(self, W_q, W_k, W_v, scale=None, mask=None):
        attention = attention_weights(W_q, W_k, scale, mask)
        if self.dropout is not None:
            attention = self.dropout(attention)
        return torch.einsum("bts,bsa->bta", attention, W_v), attention


def attention_weights(q, k, scale=None, mask=None):
    """softmax over the key axis of q k^T (divided by `scale` when given; masked positions at -inf) — the one functional
    form of the score computation both the module above and the composed path of MultiHeadAttention use."""
    scores = torch.einsum("bta,bsa->bts", q, k)
    if scale:
        scores = scores / scale
    if mask:
        scores = scores.masked_fill(mask, -np.inf)
    return torch.softmax(scores, dim=2)


class MultiHeadAttention(nn.Module):
    def __init__(self, input_dim, attention_dim=None, num_heads=1, dropout_rate=0., use_residual=True,
                 use_scale=False, layer_norm=False, align_to="input"):
        super(MultiHeadAttention, self).__init__()
        self.input_dim = input_dim
        self.attention_dim = input_dim // num_heads if attention_dim is None else attention_dim
        self.num_heads = num_heads
        self.output_dim = num_heads * self.attention_dim
        self.use_residual = use_residual
        self.align_to = align_to
        self.scale = self.attention_dim ** 0.5 if use_scale else None
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

    # -- which calls the one-launch HIP layer covers
    def _fused_ok(self, query, key, value, mask) -> bool:
        return (query.is_cuda and key is query and value is query and mask is None and self.use_residual
                and self.layer_norm is None and (self.dropout is None or not self.training)
                and (self.dot_product_attention.dropout is None or not self.training)
                and (self.W_res is None or self.align_to == "output") and query.dim() == 3
                and self._fits_lds(query.shape[1]))

    def _fits_lds(self, T) -> bool:
        """Either HIP form of the layer applies: the split form (projection GEMM + per-sample T x T core: head width
        <= 16, a sample's Q/K/V within 64 KB of LDS) or the older one-launch layer (everything, weights included, in
        the CU's 160 KB LDS).  Configurations neither covers are composed from device ops."""
        from ... import hip
        key = (T, self.input_dim, self.num_heads, self.attention_dim, self.W_res is not None)
        if getattr(self, "_fit_key", None) != key:
            self._fit_key = key
            self._fit = hip.attention_core_fits(T, self.num_heads, self.attention_dim) or hip.field_attention_fits(*key)
        return self._fit

    def _fused(self, X):
        B, T, Din = X.shape
        ws = [self.W_q.weight, self.W_k.weight, self.W_v.weight]
        if self.W_res is not None:
            ws.append(self.W_res.weight)
        from ... import hip
        W = Fh.stack_rows(ws) if all(w.dim() == 2 and w.dtype is torch.float32 for w in ws) else torch.cat(ws, dim=0)
        if hip.attention_core_fits(T, self.num_heads, self.attention_dim):
            # projections on the matrix core as one GEMM over all B*T tokens + the T x T core per sample
            return Fh.field_attention_split(X, W, T, Din, self.num_heads, self.attention_dim, self.W_res is not None,
                                            float(self.scale or 0.0))
        out = Fh.field_attention(X.reshape(B, T * Din), W, T, Din, self.num_heads,
                                 self.attention_dim, self.W_res is not None, float(self.scale or 0.0))
        return out  # [B, T, H*a]

    def _compose(self, query, key, value, mask):
        """Any configuration the HIP layer does not cover (and every CPU call): the same layer from device ops.  Heads are
        the reference's RAW view of the projections ([B, T, H*a] -> [B*H, T, a] without a per-head transpose)."""
        B, H, a = query.size(0), self.num_heads, self.attention_dim
        q, k, v = (proj(t).view(B * H, -1, a) for proj, t in ((self.W_q, query), (self.W_k, key), (self.W_v, value)))
        output, attention = self.dot_product_attention(q, k, v, self.scale, mask.repeat(H, 1, 1) if mask else mask)
        output = output.view(B, -1, self.output_dim)
        # width alignment of the skip connection: project the input up ("output") or the result back down ("input")
        residual = query
        if self.W_res is not None and self.align_to == "output":
            residual = self.W_res(query)
        elif self.W_res is not None and self.align_to == "input":
            output = self.W_res(output)
        for post in (self.dropout, (lambda o: o + residual) if self.use_residual else None, self.layer_norm):
            if post is not None:
                output = post(output)
        return output.relu(), attention

    def forward(self, query, key, value, mask=None):
        if self._fused_ok(query, key, value, mask):
            return self._fused(query), None  # attention weights are not materialised by the fused layer
        if query.is_cuda:
            from ... import hip
            hip.note_torch_path("MultiHeadAttention outside the fused layer's configurations (mask, distinct q/k/v, "
                                "layer_norm, train-mode dropout, align_to='input' or a shape the kernels' LDS budget does not hold)")
        return self._compose(query, key, value, mask)


class MultiHeadSelfAttention(MultiHeadAttention):
    def forward(self, X):
        output, _ = super(MultiHeadSelfAttention, self).forward(X, X, X)
        return output
