"""Task towers shared by the multi-task models (MMOE/OMOE/MLMMOE/ShareBottom all build the same stack:
mmoe.py:44-56, omoe.py:44-56, mlmmoe.py:52-64, sharebottom.py:40-52):

    task_{i}_dnn = ModuleList[ctr_hidden_j Linear, ctr_batchnorm_j BatchNorm1d, ctr_dropout_j Dropout]*  +
                   task_last_layer Linear(.,1) + task_sigmoid
    loss = sum_t w_t * BCE(p_t (+ p_eps), y_t),   w_t = 1/T

HIP execution: Linear on the fp32-MFMA kernel, BatchNorm1d on rp_batchnorm_*, the final sigmoid fused with
the BCE in rp_sigmoid_bce_*; Dropout on rp_dropout_* (Philox mask) when training with p > 0.
"""
import numpy as np
from torch import nn

from ... import functional as Fh


def build_towers(model: nn.Module, num_task: int, in_dim: int, hidden_dim, dropouts):
    """Registers task_{i}_dnn on `model`, same module names and creation order as the reference."""
    for i in range(num_task):
        tower = nn.ModuleList()
        setattr(model, 'task_{}_dnn'.format(i + 1), tower)
        hid_dim = [in_dim] + list(hidden_dim)
        for j in range(len(hid_dim) - 1):
            tower.add_module('ctr_hidden_{}'.format(j), nn.Linear(hid_dim[j], hid_dim[j + 1]))
            tower.add_module('ctr_batchnorm_{}'.format(j), nn.BatchNorm1d(hid_dim[j + 1]))
            tower.add_module('ctr_dropout_{}'.format(j), nn.Dropout(dropouts[j]))
        tower.add_module('task_last_layer', nn.Linear(hid_dim[-1], 1))
        tower.add_module('task_sigmoid', nn.Sigmoid())


def weighted_bce(task_outputs, data, num_task: int, p_eps: float = 0.0, weight=None):
    """`loss()` of the multi-task models on probabilities (CPU tensors: torch; HIP tensors: the loss kernel)."""
    if weight is None:
        weight = np.ones(num_task) / num_task
    total = 0
    if len(task_outputs) and all(p.is_cuda for p in task_outputs):
        labels = [data[f'task{i + 1}_label'].float() for i in range(len(task_outputs))]
        return Fh.sigmoid_bce_multi(list(task_outputs), labels, weight, apply_sigmoid=False, p_eps=p_eps)[1]
    for i, p in enumerate(task_outputs):
        y = data[f'task{i + 1}_label']
        if p.is_cuda:
            _, l_i = Fh.sigmoid_bce([p], y.float(), apply_sigmoid=False, p_eps=p_eps, weight=float(weight[i]))
            total = total + l_i
        else:
            pp = p.squeeze(-1) + p_eps if p_eps else p.squeeze(-1)
            total = total + weight[i] * nn.functional.binary_cross_entropy(pp, y)
    return total


def run_towers(model: nn.Module, inputs, data, is_training: bool, p_eps: float = 0.0):
    """inputs[i] = tower i's input [B, in_dim] -> {'task{i}_pred': [B,1], 'loss'}."""
    T = model.num_task
    output_dict = dict()
    if not inputs[0].is_cuda:
        task_outputs = []
        for i in range(T):
            x_t = inputs[i]
            for mod in getattr(model, 'task_{}_dnn'.format(i + 1)):
                x_t = mod(x_t)
            task_outputs.append(x_t)
            output_dict[f'task{i + 1}_pred'] = x_t
        if is_training:
            output_dict['loss'] = model.loss(task_outputs, data)
        return output_dict
    logits = []
    for i in range(T):
        x_t = inputs[i]
        for mod in getattr(model, 'task_{}_dnn'.format(i + 1)):
            if isinstance(mod, nn.Linear):
                x_t = Fh.linear_act(x_t, mod.weight, mod.bias, Fh.ACT_NONE)
            elif isinstance(mod, nn.Sigmoid):
                break  # fused into the loss / prediction kernel below
            elif isinstance(mod, nn.BatchNorm1d):
                x_t = Fh.batch_norm(x_t, mod)
            elif isinstance(mod, nn.Dropout):
                if model.training and 0 < mod.p < 1:
                    x_t = Fh.dropout(x_t, mod.p)  # rp_dropout_*
                elif model.training and mod.p >= 1:
                    x_t = mod(x_t)
            else:
                from ... import hip
                hip.note_torch_path(f"{type(mod).__name__} inside a task tower")
                x_t = mod(x_t)
        if is_training:
            logits.append(x_t)
        else:
            output_dict[f'task{i + 1}_pred'] = Fh.sigmoid_sum([x_t])
    if is_training:
        # one loss scalar for all tasks: every task's launch adds its weighted mean to it, in task order
        labels = [data[f'task{i + 1}_label'].float() for i in range(T)]
        preds, output_dict['loss'] = Fh.sigmoid_bce_multi(logits, labels, [1.0 / T] * T, apply_sigmoid=True, p_eps=p_eps)
        for i in range(T):
            output_dict[f'task{i + 1}_pred'] = preds[i]
    return output_dict
