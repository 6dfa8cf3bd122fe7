"""BaseDataset — counterpart of rec_pangu/dataset/base_dataset.py:14-133 (input format of the hot path).

schema + DataFrame -> enc_dict -> dict-of-[B] tensors: sparse ids int64 in [0, vocab] (vocab = OOV
row), dense (x-min)/(max-min+1e-5) float32, 'label' float32.

Kept from the reference on purpose (results must match on the same inputs):
  * ids go through float32 (torch.Tensor(...).long(), base_dataset.py:103) -> exact only below 2^24 (B5);
  * a column is cast to str only while the enc_dict is being BUILT (base_dataset.py:58); a dataset made
    with a given enc_dict looks raw values up, so numeric-typed categorical columns of valid/test frames
    all map to the OOV id (B8, pinned by tests/golden/dataset.npz).
Changed on purpose: column order is the schema's order, de-duplicated, instead of list(set(...)) which
depends on PYTHONHASHSEED (B4); pass the reference's enc_dict to reproduce a reference run's field order.
The per-row pandas .apply is replaced by a vectorised map (same result).
"""
from typing import Dict

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset


def _unique_in_order(cols):
    return list(dict.fromkeys(cols))


class BaseDataset(Dataset):
    def __init__(self, config: dict, df: pd.DataFrame, enc_dict: Dict[str, dict] = None):
        self.config = config
        self.enc_dict = enc_dict
        self.df = df.rename(columns={self.config['label_col']: 'label'}) if isinstance(self.config['label_col'], str) \
            else df.copy()
        self.dense_cols = _unique_in_order(self.config['dense_cols'])
        self.sparse_cols = _unique_in_order(self.config['sparse_cols'])
        self.feature_name = self.dense_cols + self.sparse_cols
        if self.enc_dict is None:
            self.get_enc_dict()
        self.enc_data()

    def get_enc_dict(self) -> Dict[str, dict]:
        self.enc_dict = {c: dict() for c in self.dense_cols + self.sparse_cols}
        for f in self.sparse_cols:
            self.df[f] = self.df[f].astype('str')
            cats = sorted(self.df[f].unique())
            self.enc_dict[f] = dict(zip(cats, range(len(cats))))
            self.enc_dict[f]['vocab_size'] = len(cats)
        for f in self.dense_cols:
            self.enc_dict[f]['min'] = self.df[f].min()
            self.enc_dict[f]['max'] = self.df[f].max()
        return self.enc_dict

    def enc_dense_data(self, col: str):
        lo, hi = self.enc_dict[col]['min'], self.enc_dict[col]['max']
        return (self.df[col] - lo) / (hi - lo + 1e-5)

    def enc_sparse_data(self, col: str):
        mapping = self.enc_dict[col]
        oov = mapping['vocab_size']
        values = self.df[col].to_numpy()
        return np.fromiter((mapping.get(v, oov) for v in values.tolist()), dtype=np.int64, count=len(values))

    def enc_data(self):
        self.data_dict = {}
        for col in self.dense_cols:
            self.data_dict[col] = torch.Tensor(np.array(self.enc_dense_data(col)))
        for col in self.sparse_cols:
            self.data_dict[col] = torch.Tensor(np.array(self.enc_sparse_data(col))).long()
        if 'label' in self.df.columns:
            self._label = torch.Tensor(self.df['label'].to_numpy(dtype=np.float32))

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        data = {col: self.data_dict[col][index] for col in self.feature_name}
        if 'label' in self.df.columns:
            data['label'] = self._label[index]
        return data

    def _label_columns(self):
        """label tensors by the key __getitem__ gives them (DeviceBatchLoader moves whole columns)."""
        return {'label': self._label} if 'label' in self.df.columns else {}

    def __len__(self) -> int:
        return len(self.df)
