"""MultiTaskDataset — the INTENDED behaviour of rec_pangu/dataset/multi_task_dataset.py:14-80.

The reference class cannot be constructed at v0.4.1 (it calls an undefined self.data()),
so there is nothing to be bit-compatible with; this provides what its __getitem__ documents: the
BaseDataset features plus one float label per task under `task{i}_label`, label_col being a list.
"""
from typing import Dict

import numpy as np
import pandas as pd
import torch

from .base_dataset import BaseDataset


class MultiTaskDataset(BaseDataset):
    def __init__(self, config: dict, df: pd.DataFrame, enc_dict: Dict[str, dict] = None):
        renames = {col: f'task{i + 1}_label' for i, col in enumerate(config['label_col'])}
        super().__init__(dict(config, label_col=list(config['label_col'])), df.rename(columns=renames), enc_dict)
        self._task_labels = {name: torch.Tensor(self.df[name].to_numpy(dtype=np.float32))
                             for name in renames.values() if name in self.df.columns}

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        data = {col: self.data_dict[col][index] for col in self.feature_name}
        for name, t in self._task_labels.items():
            data[name] = t[index]
        return data

    def _label_columns(self):
        return dict(self._task_labels)
