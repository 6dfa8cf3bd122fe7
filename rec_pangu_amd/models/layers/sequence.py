"""Sequence poolings of the reference (rec_pangu/models/layers/sequence.py:13-59) as drop-in modules.

On a HIP tensor they run rp_seq_pool_fwd / _bwd (one read of the [B, L, D] tensor); on the CPU the reference's own
formulas.  When the pooled tensor comes straight from a `_seq` lookup, `EmbeddingLayer.lookup_pooled` fuses lookup and
pooling and the [B, L, D] tensor never exists.
"""
import torch
from torch import nn

from ... import functional as Fh


class MaskedAveragePooling(nn.Module):
    """sum over dim 1 / (number of non-zero elements per (sample, column) + 1e-16) — sequence.py:30-36"""

    def forward(self, embedding_matrix: torch.Tensor) -> torch.Tensor:
        if embedding_matrix.is_cuda and embedding_matrix.dim() == 3:
            return Fh.seq_pool(embedding_matrix, "average")
        nonzero = embedding_matrix.ne(0).sum(dim=1).to(embedding_matrix.dtype)  # per (sample, column), like the reference
        return embedding_matrix.sum(dim=1) / (nonzero + 1e-16)


class MaskedSumPooling(nn.Module):
    """sum over dim 1 (padding rows count unless they are zero) — sequence.py:58-59"""

    def forward(self, embedding_matrix: torch.Tensor) -> torch.Tensor:
        if embedding_matrix.is_cuda and embedding_matrix.dim() == 3:
            return Fh.seq_pool(embedding_matrix, "sum")
        return embedding_matrix.sum(dim=1)
