from .base_dataset import BaseDataset
from .multi_task_dataset import MultiTaskDataset
from .process_data import get_dataloader, get_single_dataloader
from .device_loader import DeviceBatchLoader, PinnedBatchLoader

__all__ = ["BaseDataset", "MultiTaskDataset", "get_dataloader", "get_single_dataloader", "DeviceBatchLoader",
           "PinnedBatchLoader"]
