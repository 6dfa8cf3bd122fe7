"""DeviceBatchLoader — a batch iterator over an encoded dataset whose columns already live on the device.

The reference feeds the model through torch's DataLoader(num_workers=0) over a per-sample `__getitem__`
(`process_data.py:18`, `base_dataset.py:105-120`): a Python call and a dict per SAMPLE, then a collate — a few
hundred thousand samples per second at best, two orders of magnitude below what the HIP path consumes
(SURVEY.md §8f rank 3).  The encoded columns of a BaseDataset / MultiTaskDataset are plain tensors
(`data_dict`, labels); this loader moves them to the device ONCE and yields each batch as a dict of slices
(shuffle = one device `randperm` + one gather per column per epoch), i.e. exactly the dicts the DataLoader would
have produced, in the same order when `shuffle=False`.  RankTrainer / BenchmarkTrainer accept it wherever they
accept a DataLoader (`.dataset`, `len()`, iteration).
"""
from typing import Dict, Iterator, Optional

import torch


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size: int, shuffle: bool = False, device: Optional[torch.device] = None,
                 drop_last: bool = False, generator: Optional[torch.Generator] = None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last, self.generator = shuffle, drop_last, generator
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        n = len(dataset)
        cols = {k: dataset.data_dict[k] for k in dataset.feature_name}
        cols.update(dataset._label_columns())  # 'label' / 'task{i}_label': what __getitem__ adds to the features
        self.columns: Dict[str, torch.Tensor] = {k: v.to(self.device) for k, v in cols.items()}
        self.n = n

    def __len__(self) -> int:
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        cols = self.columns
        if self.shuffle:
            perm = torch.randperm(self.n, device=self.device, generator=self.generator)
            cols = {k: v[perm] for k, v in cols.items()}
        for i in range(len(self)):
            lo = i * self.batch_size
            yield {k: v[lo:lo + self.batch_size] for k, v in cols.items()}
