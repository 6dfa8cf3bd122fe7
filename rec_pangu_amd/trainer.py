"""RankTrainer — counterpart of rec_pangu/trainer.py:23-236 (single- and multi-task ranking).

Same constructor and method signatures, same epoch loop (train -> validate -> per-epoch checkpoint
`model_e_{i}.pth` -> optional early stopping with `model_best.pth`), same checkpoint layout
({'model': state_dict[, 'enc_dict': ...]}), same Adam hyper-parameters.  The optimiser is created
after the model has been moved (the reference creates it before `.to(device)`, trainer.py:75-76; the
state is lazy either way) so that a HIP-resident model gets the fused HIP Adam.
"""
import logging
import os
from typing import Optional

import torch
import torch.utils.data as D
from torch.optim import lr_scheduler

from .dataset import BaseDataset, MultiTaskDataset
from .model_pipeline import train_model, test_model
from .optim import make_adam

logger = logging.getLogger("rec_pangu_amd")


class RankTrainer:
    def __init__(self, num_task: int = 1, wandb_config: dict = None, model_ckpt_dir: str = './model_ckpt'):
        self.num_task = num_task
        self.wandb_config = wandb_config
        self.model_ckpt_dir = model_ckpt_dir
        self.use_wandb = self.wandb_config is not None
        if self.use_wandb:
            import wandb
            wandb.login(key=self.wandb_config['key'])
            self.wandb_config.pop('key')

    def fit(self, model, train_loader, valid_loader: Optional = None, epoch: int = 10, lr: float = 1e-3,
            device: torch.device = torch.device('cpu'), use_earlystopping: bool = False, max_patience: int = 999,
            monitor_metric: Optional[str] = None, lr_scheduler_type: str = "",
            scheduler_params: Optional[dict] = {}, use_hip_graph: bool = False):
        """The reference's signature (trainer.py:51-54) plus `use_hip_graph`: replay the training steps from a captured
        hipGraph (graph_step.GraphedTrainStep: HIP models without active dropout; out-of-range ids are then reported at
        the end of the epoch instead of at the offending step)."""
        if self.use_wandb:
            import wandb
            wandb.init(**self.wandb_config)
        model = model.to(device)
        optimizer = make_adam(model, lr)
        graphed_step = None
        if use_hip_graph:
            from .graph_step import GraphedTrainStep
            for m in model.modules():
                if hasattr(m, "check_indices"):
                    m.check_indices = "deferred"
            graphed_step = GraphedTrainStep(model, optimizer)

        schedulers = {'StepLR': lr_scheduler.StepLR, 'ExponentialLR': lr_scheduler.ExponentialLR,
                      'CosineAnnealingLR': lr_scheduler.CosineAnnealingLR}
        if lr_scheduler_type not in schedulers and lr_scheduler_type != "":
            raise ValueError('Unknown scheduler type: {}'.format(lr_scheduler_type))
        scheduler = schedulers[lr_scheduler_type](optimizer, **scheduler_params) if lr_scheduler_type else None

        logger.info('Model Starting Training ')
        best_epoch, best_metric = -1, -1
        valid_metric = None
        for i in range(1, epoch + 1):
            train_metric = train_model(model, train_loader, optimizer=optimizer, device=device,
                                       num_task=self.num_task, use_wandb=self.use_wandb, graphed_step=graphed_step)
            if scheduler is not None:
                scheduler.step()
                logger.info(f"Epoch {i} LR:{[round(x, 6) for x in scheduler.get_last_lr()]}")
            logger.info(f"Train Metric:{train_metric}")
            if valid_loader is not None:
                valid_metric = test_model(model, valid_loader, device, num_task=self.num_task)
                self.save_train_model(model, self.model_ckpt_dir, f'e_{i}')
                if self.use_wandb:
                    import wandb
                    wandb.log(valid_metric)
                if use_earlystopping:
                    assert monitor_metric in valid_metric.keys(), \
                        f'{monitor_metric} not in Valid Metric {valid_metric.keys()}'
                    if valid_metric[monitor_metric] > best_metric:
                        best_epoch, best_metric = i, valid_metric[monitor_metric]
                        self.save_train_model(model, self.model_ckpt_dir, 'best')
                    if i - best_epoch >= max_patience:
                        logger.info(f"EarlyStopping at the Epoch {i} Valid Metric:{valid_metric}")
                        break
                logger.info(f"Valid Metric:{valid_metric}")
        if self.use_wandb:
            import wandb
            wandb.finish()
        return valid_metric

    # ---- checkpoints: reference layout (trainer.py:124-164) -------------------------------------
    @staticmethod
    def _save(payload, model_ckpt_dir, filename):
        os.makedirs(model_ckpt_dir, exist_ok=True, mode=0o777)
        torch.save(payload, os.path.join(model_ckpt_dir, filename))
        logger.info(f'Model Saved to {model_ckpt_dir}')

    @staticmethod
    def _is_sharded(model) -> bool:
        return any(type(m).__name__ == "ShardedEmbeddingLayer" for m in model.modules())

    def _save_sharded(self, model, enc_dict, model_ckpt_dir, filename):
        """Row-sharded tables (rec_pangu_amd.sharded): every rank writes its shard, rank 0 merges them into the same
        {'model': <reference state_dict keys>[, 'enc_dict']} file the reference writes (rec_pangu_amd.checkpoint)."""
        from .checkpoint import save_checkpoint
        save_checkpoint(model, enc_dict, model_ckpt_dir, filename=filename)
        logger.info(f'Model Saved to {model_ckpt_dir}')

    def save_model(self, model, model_ckpt_dir: str):
        if self._is_sharded(model):
            return self._save_sharded(model, None, model_ckpt_dir, 'model.pth')
        self._save({'model': model.state_dict()}, model_ckpt_dir, 'model.pth')

    def save_all(self, model, enc_dict: dict, model_ckpt_dir: str):
        if self._is_sharded(model):
            return self._save_sharded(model, enc_dict, model_ckpt_dir, 'model.pth')
        self._save({'model': model.state_dict(), 'enc_dict': enc_dict}, model_ckpt_dir, 'model.pth')

    def save_train_model(self, model, model_ckpt_dir: str, model_str: str):
        if self._is_sharded(model):
            return self._save_sharded(model, None, model_ckpt_dir, f'model_{model_str}.pth')
        self._save({'model': model.state_dict()}, model_ckpt_dir, f'model_{model_str}.pth')

    # ---- evaluation / inference (trainer.py:166-236) --------------------------------------------
    def evaluate_model(self, model, test_loader, device: torch.device = torch.device('cpu')):
        test_metric = test_model(model, test_loader, device, num_task=self.num_task)
        logger.info(f"Test Metric:{test_metric}")
        return test_metric

    def predict_dataloader(self, model, test_loader, device: torch.device = torch.device('cpu')):
        model.eval()
        chunks = [[] for _ in range(self.num_task)]
        for data in test_loader:
            for key in data.keys():
                data[key] = data[key].to(device)
            output = model(data, is_training=False)
            for i in range(self.num_task):
                key = 'pred' if self.num_task == 1 else f'task{i + 1}_pred'
                chunks[i].append(output[key].detach().reshape(-1))
        out = [list(torch.cat(c).cpu().numpy()) if c else [] for c in chunks]
        return out[0] if self.num_task == 1 else out

    def predict_dataframe(self, model, test_df, enc_dict: dict, schema: dict,
                          device: torch.device = torch.device('cpu'), batch_size: int = 1024):
        if schema['task_type'] == 'ranking':
            test_dataset = BaseDataset(schema, test_df, enc_dict=enc_dict)
        elif schema['task_type'] == 'multitask':
            test_dataset = MultiTaskDataset(schema, test_df, enc_dict=enc_dict)
        test_loader = D.DataLoader(test_dataset, batch_size=batch_size, shuffle=False, num_workers=0)
        return self.predict_dataloader(model, test_loader, device=device)
