"""enc_dict helpers of the hot path (reference: rec_pangu/models/utils.py:122-170)."""
from typing import Dict, List, Tuple

import torch


def dense_columns(enc_dict: Dict) -> List[str]:
    """Dense column order = enc_dict key order of the entries that carry 'min' (utils.py:133-135)."""
    return [col for col in enc_dict.keys() if "min" in enc_dict[col].keys()]


def get_linear_input(enc_dict: Dict, data: Dict) -> torch.Tensor:
    """[B, ND] stack of the dense columns (utils.py:122-137)."""
    return torch.stack([data[col] for col in dense_columns(enc_dict)], axis=1)


def get_feature_num(enc_dict: Dict) -> Tuple[int, int]:
    """(num_sparse, num_dense) (utils.py:154-170): 'min' wins over 'vocab_size' when both are present."""
    num_sparse = num_dense = 0
    for col in enc_dict.keys():
        if "min" in enc_dict[col].keys():
            num_dense += 1
        elif "vocab_size" in enc_dict[col].keys():
            num_sparse += 1
    return num_sparse, num_dense


def get_dnn_input_dim(enc_dict: Dict, embedding_dim: int) -> int:
    num_sparse, num_dense = get_feature_num(enc_dict)
    return num_sparse * embedding_dim + num_dense
