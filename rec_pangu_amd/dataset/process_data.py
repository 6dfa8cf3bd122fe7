"""get_dataloader — counterpart of rec_pangu/dataset/process_data.py:12-77 for 'ranking' and 'multitask'."""
import torch.utils.data as D

from .base_dataset import BaseDataset
from .multi_task_dataset import MultiTaskDataset


def _loaders(cls, train_df, valid_df, test_df, schema, batch_size):
    train_dataset = cls(schema, train_df)
    enc_dict = train_dataset.enc_dict
    valid_dataset = cls(schema, valid_df, enc_dict=enc_dict)
    test_dataset = cls(schema, test_df, enc_dict=enc_dict)
    mk = lambda ds, shuffle: D.DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0, pin_memory=False)
    return mk(train_dataset, True), mk(valid_dataset, False), mk(test_dataset, False), enc_dict


def get_base_dataloader(train_df, valid_df, test_df, schema, batch_size=512 * 3):
    return _loaders(BaseDataset, train_df, valid_df, test_df, schema, batch_size)


def get_multi_task_dataloader(train_df, valid_df, test_df, schema, batch_size=512 * 3):
    return _loaders(MultiTaskDataset, train_df, valid_df, test_df, schema, batch_size)


def get_dataloader(train_df, valid_df, test_df, schema, batch_size=512 * 3):
    if schema['task_type'] == 'ranking':
        return get_base_dataloader(train_df, valid_df, test_df, schema, batch_size=batch_size)
    elif schema['task_type'] == 'multitask':
        return get_multi_task_dataloader(train_df, valid_df, test_df, schema, batch_size=batch_size)
    raise Exception(f"""task_type:{schema['task_type']} must be in ['ranking','multitask']""")


def get_single_dataloader(test_df, schema, enc_dict, batch_size=512, num_workers=0):
    cls = MultiTaskDataset if schema['task_type'] == 'multitask' else BaseDataset
    return D.DataLoader(cls(schema, test_df, enc_dict=enc_dict), batch_size=batch_size, shuffle=False,
                        num_workers=num_workers)
