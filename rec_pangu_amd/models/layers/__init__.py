from .activation import Dice, get_activation
from .embedding import EmbeddingLayer
from .deep import MLP
from .shallow import LR_Layer
from .interaction import (InnerProductLayer, FM_Layer, CrossInteractionLayer, CrossNet,
                          CompressedInteractionNet)
from .attention import ScaledDotProductAttention, MultiHeadAttention, MultiHeadSelfAttention
from .sequence import MaskedAveragePooling, MaskedSumPooling

__all__ = ["Dice", "get_activation", "EmbeddingLayer", "MLP", "LR_Layer", "InnerProductLayer", "FM_Layer",
           "CrossInteractionLayer", "CrossNet", "CompressedInteractionNet", "ScaledDotProductAttention",
           "MultiHeadAttention", "MultiHeadSelfAttention", "MaskedAveragePooling", "MaskedSumPooling"]
