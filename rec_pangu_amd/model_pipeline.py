"""train_model / test_model — counterparts of rec_pangu/model_pipeline.py:17-219 (ranking + multi-task).

Same call order per batch (forward -> loss.backward() -> optimizer.step() -> model.zero_grad()),
same returned keys (`train_roc_auc_score`, `train_log_loss`, `roc_auc_score`, `log_loss`,
`train_task{i}_*`, `test_task{i}_*`), same 4-decimal rounding and log-loss clipping (eps=1e-7).
Two things differ, neither changes a returned number: predictions stay on the device until the end of
the epoch (the reference does a .cpu() round trip every iteration, model_pipeline.py:60-61), and the
running "last 1000" AUC is only computed on the iterations that log it.
"""
import logging
import time
from typing import List

import numpy as np
import torch
from sklearn.metrics import roc_auc_score
from sklearn.metrics import log_loss as _sk_log_loss

logger = logging.getLogger("rec_pangu_amd")
SUPPORTED_METRICS = ['roc_auc_score', 'log_loss']


def log_loss(y_true, y_pred, eps=1e-7):
    """sklearn.metrics.log_loss with the clipping the reference asks for (`eps=1e-7`, removed in sklearn 1.5)."""
    y_pred = np.clip(np.asarray(y_pred, dtype=np.float64), eps, 1 - eps)
    return _sk_log_loss(y_true, y_pred)


def _metric(name, labels, preds):
    assert name in SUPPORTED_METRICS, 'metric :{} not supported! metric must be in {}'.format(name, SUPPORTED_METRICS)
    if name == 'log_loss':
        return round(log_loss(labels, preds, eps=1e-7), 4)
    return round(roc_auc_score(labels, preds), 4)


def _to_host(chunks):
    if not chunks:
        return np.zeros((0,), dtype=np.float32)
    return torch.cat([c.reshape(-1) for c in chunks]).cpu().numpy()


def _move(data, device):
    for key in data.keys():
        data[key] = data[key].to(device)
    return data


def _check_indices(model):
    for m in model.modules():
        if hasattr(m, "raise_if_bad_index"):
            m.raise_if_bad_index()


def _one_ahead(loader, device, model, pairs: bool = False):
    """The loader's batches in order, each yielded after the NEXT one has been moved to the device and announced to the
    model (BaseModel.prefetch: its row sort overlaps the step in flight).  Same batches, same order, same RNG draws.
    pairs: yield (batch, next batch or None) and leave the announcing to the caller (a GraphedTrainStep)."""
    announce = getattr(model, "prefetch", None)
    if hasattr(loader, "hold"):
        loader.hold = 2  # loaders that recycle device buffers (PinnedBatchLoader): two batches are alive at once
    it = iter(loader)
    try:
        cur = _move(next(it), device)
    except StopIteration:
        return
    for nxt in it:
        nxt = _move(nxt, device)
        if announce is not None and not pairs:
            announce(nxt)
        yield (cur, nxt) if pairs else cur
        cur = nxt
    yield (cur, None) if pairs else cur


def train_model(model, train_loader, optimizer, device, metric_list: List[str] = ['roc_auc_score', 'log_loss'],
                num_task: int = 1, use_wandb: bool = False, log_rounds: int = 100, graphed_step=None) -> dict:
    """graphed_step: a rec_pangu_amd.graph_step.GraphedTrainStep over (model, optimizer): the steps are then replays of
    a captured hipGraph (same launches, bit-identical results; batches of another shape — the last one of an epoch —
    fall back to the eager step)."""
    if graphed_step is not None:
        return _train_model_graphed(model, train_loader, device, metric_list, num_task, log_rounds, graphed_step)
    model.train()
    max_iter = int(train_loader.dataset.__len__() / train_loader.batch_size)
    tasks = range(num_task)
    preds = [[] for _ in tasks]
    labels = [[] for _ in tasks]
    start_time = time.time()
    for idx, data in enumerate(_one_ahead(train_loader, device, model)):
        output = model(data)
        loss = output['loss']
        loss.backward()
        optimizer.step()
        model.zero_grad()
        for i in tasks:
            pk, lk = ('pred', 'label') if num_task == 1 else (f'task{i + 1}_pred', f'task{i + 1}_label')
            preds[i].append(output[pk].detach())
            labels[i].append(data[lk].detach())
        if use_wandb:
            import wandb
            wandb.log({'train_loss': loss.item()})
        if idx % log_rounds == 0:
            elapsed = time.time() - start_time
            remaining = round(((elapsed / (idx + 1)) * (max_iter - idx + 1)) / 60, 2)
            logger.info(f'Iter {idx}/{max_iter} Remaining time:{remaining} min Loss:{round(float(loss.detach()), 4)}')
    _check_indices(model)
    res = dict()
    for i in tasks:
        y, p = _to_host(labels[i]), _to_host(preds[i])
        for metric in metric_list:
            key = f'train_{metric}' if num_task == 1 else f'train_task{i + 1}_{metric}'
            res[key] = _metric(metric, y, p)
    return res


def _train_model_graphed(model, train_loader, device, metric_list, num_task, log_rounds, graphed_step) -> dict:
    model.train()
    tasks = range(num_task)
    preds = [[] for _ in tasks]
    labels = [[] for _ in tasks]
    for idx, (data, nxt) in enumerate(_one_ahead(train_loader, device, model, pairs=True)):
        output = graphed_step(data, nxt)
        for i in tasks:
            pk, lk = ('pred', 'label') if num_task == 1 else (f'task{i + 1}_pred', f'task{i + 1}_label')
            preds[i].append(output[pk].detach().clone())  # (a static tensor of the graph: the next replays overwrite it)
            labels[i].append(data[lk].detach())
        if idx % log_rounds == 0:
            logger.info(f'Iter {idx} Loss:{round(float(output["loss"].detach()), 4)}')
    _check_indices(model)
    res = dict()
    for i in tasks:
        y, p = _to_host(labels[i]), _to_host(preds[i])
        for metric in metric_list:
            key = f'train_{metric}' if num_task == 1 else f'train_task{i + 1}_{metric}'
            res[key] = _metric(metric, y, p)
    return res


def test_model(model, test_loader, device, metric_list: List[str] = ['roc_auc_score', 'log_loss'],
               num_task: int = 1) -> dict:
    model.eval()
    tasks = range(num_task)
    preds = [[] for _ in tasks]
    labels = [[] for _ in tasks]
    for data in test_loader:
        data = _move(data, device)
        output = model(data)  # is_training left at its default, like the reference (model_pipeline.py:164)
        for i in tasks:
            pk, lk = ('pred', 'label') if num_task == 1 else (f'task{i + 1}_pred', f'task{i + 1}_label')
            preds[i].append(output[pk].detach())
            labels[i].append(data[lk].detach())
    _check_indices(model)
    res = dict()
    for i in tasks:
        y, p = _to_host(labels[i]), _to_host(preds[i])
        for metric in metric_list:
            key = metric if num_task == 1 else f'test_task{i + 1}_{metric}'
            res[key] = _metric(metric, y, p)
    return res


test_model.__test__ = False  # not a pytest test
