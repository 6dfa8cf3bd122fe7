"""BenchmarkTrainer — counterpart of rec_pangu/benchmark_trainer.py:18-95.

Trains each named model with RankTrainer.fit, times fit/evaluate with the wall clock, saves
{'model','enc_dict'} under ckpt_root/<name>/ and rewrites the CSV after every model with the columns
`model_name, train_model_time, test_model_time, <valid metrics>, <test metrics>` (milliseconds).
Model names resolve through an explicit registry of the hot-path models instead of eval().
"""
import inspect
import logging
import os
import time
from typing import Dict, List, Optional

import pandas as pd
import torch
from torch.utils.data import DataLoader

from .models import ranking as _ranking, multi_task as _multi_task
from .trainer import RankTrainer

logger = logging.getLogger("rec_pangu_amd")

MODEL_REGISTRY = {name: getattr(mod, name) for mod in (_ranking, _multi_task) for name in mod.__all__}


class BenchmarkTrainer:
    def __init__(self, num_task: int = 1, model_list: Optional[List[str]] = None,
                 benchmark_res_path: Optional[str] = None, ckpt_root: str = './benchmark_ckpt') -> None:
        self.num_task = num_task
        self.model_list = model_list
        self.benchmark_res_df = pd.DataFrame()
        self.benchmark_res_path = benchmark_res_path
        self.ckpt_root = ckpt_root

    def run(self, train_loader: DataLoader, enc_dict: Dict[str, int], valid_loader: Optional[DataLoader] = None,
            test_loader: Optional[DataLoader] = None, epoch: int = 10, lr: float = 1e-3,
            device: torch.device = torch.device('cpu')) -> None:
        rows = []
        for model_name in self.model_list:
            logger.info(f'Start Training Model: {model_name}')
            if model_name not in MODEL_REGISTRY:
                raise NameError(f"name '{model_name}' is not defined")  # what the reference's eval() raises
            model_class = MODEL_REGISTRY[model_name]
            # benchmark_trainer.py:67-68 passes device= to every multi-task model although ShareBottom's constructor
            # has no such argument (TypeError in the reference): pass it only where it is accepted
            kw = {"device": device} if (self.num_task > 1 and "device" in
                                        inspect.signature(model_class.__init__).parameters) else {}
            model = model_class(enc_dict=enc_dict, **kw)
            ckpt_dir = os.path.join(self.ckpt_root, model_name)
            trainer = RankTrainer(num_task=self.num_task, model_ckpt_dir=ckpt_dir)

            t0 = time.time()
            valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=epoch, lr=lr, device=device)
            train_ms = (time.time() - t0) * 1000
            t0 = time.time()
            test_metric = trainer.evaluate_model(model, test_loader, device=device) if test_loader is not None else {}
            test_ms = (time.time() - t0) * 1000
            trainer.save_all(model, enc_dict, ckpt_dir)

            log_dict = {'model_name': model_name, 'train_model_time': train_ms, 'test_model_time': test_ms}
            logger.info(f'Model {model_name} Training Log :{log_dict}')
            log_dict.update(valid_metric or {})
            log_dict.update(test_metric)
            rows.append(log_dict)
            self.benchmark_res_df = pd.DataFrame(rows)
            self.benchmark_res_df.to_csv(self.benchmark_res_path, index=False)
