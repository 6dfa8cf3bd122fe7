"""CPU oracle for the rec_pangu ranking hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module;
nothing under rec_pangu_amd/ does (tests/test_no_oracle_in_product.py enforces it).

It is a functional restatement (explicit weight dicts in, tensors out; plain torch CPU fp32 ops,
because ATen fp32 *is* the reference's arithmetic) of the reference algorithm for
the path BASELINE.json names.  Every function cites the reference file:line it follows
(paths relative to the upstream repo root).  Backward passes come from torch.autograd on
these forward restatements, exactly as the reference gets its own.

Pinning: the reference ships no tests, so the pin is tests/golden/*.npz — outputs
of the reference itself, imported and run in the build container by tests/golden/make_golden.py.
tests/test_oracle_golden.py checks every function below against those vectors.

Weights use the reference's state_dict key names so a reference checkpoint can be
fed in unchanged.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# enc_dict helpers  (rec_pangu/models/utils.py:122-170)
# ------------------------------------------------------------------------------------------------
def sparse_fields(enc_dict: Dict[str, dict]) -> List[str]:
    """Field order = enc_dict key order of entries having 'vocab_size' (layers/embedding.py:28-30)."""
    return [c for c in enc_dict.keys() if "vocab_size" in enc_dict[c]]


def dense_fields(enc_dict: Dict[str, dict]) -> List[str]:
    """Dense column order = enc_dict key order of entries having 'min' (models/utils.py:133-135)."""
    return [c for c in enc_dict.keys() if "min" in enc_dict[c]]


def get_linear_input(enc_dict, data) -> Tensor:
    """models/utils.py:122-137: stack dense columns -> [B, ND]."""
    return torch.stack([data[c] for c in dense_fields(enc_dict)], dim=1)


# ------------------------------------------------------------------------------------------------
# embedding  (rec_pangu/models/layers/embedding.py:49-71)
# ------------------------------------------------------------------------------------------------
def embedding_all(tables: Dict[str, Tensor], enc_dict, data) -> Tensor:
    """embedding.py:58-63: one row per field per sample, stacked to [B, F, D]; bag size 1, no pooling."""
    outs = []
    for col in sparse_fields(enc_dict):
        idx = data[col].long().view(-1)
        w = tables[col]
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= w.shape[0]):
            raise IndexError("index out of range in self")  # what nn.Embedding raises on CPU
        outs.append(w[idx])
    return torch.stack(outs, dim=1)


def embedding_by_name(tables: Dict[str, Tensor], data, name: str) -> Tensor:
    """embedding.py:64-71: single field -> [B,1,D]; '<col>_seq' -> [B,L,D] from table <col>."""
    if "seq" in name:
        return tables[name.replace("_seq", "")][data[name].long()]
    return tables[name][data[name].long().view(-1, 1)]


def seq_pool(emb_seq: Tensor, mode: str) -> Tensor:
    """rec_pangu/models/layers/sequence.py:58-59 (MaskedSumPooling: torch.sum(dim=1)) and :30-36
    (MaskedAveragePooling: sum / (count of non-zero ELEMENTS per (sample, column) + 1e-16)) on [B, L, D]."""
    s = torch.sum(emb_seq, dim=1)
    if mode == "sum":
        return s
    non_padding_length = (emb_seq != 0).sum(dim=1)
    return s / (non_padding_length.float() + 1e-16)


def embedding_seq_pooled(tables: Dict[str, Tensor], data, name: str, mode: str) -> Tensor:
    """the `_seq` lookup (embedding.py:64-67) followed by one of the two poolings above -> [B, D]"""
    return seq_pool(embedding_by_name(tables, data, name), mode)


def embedding_bags_pooled(table: Tensor, ids: Tensor, offsets: Tensor, mode: str) -> Tensor:
    """the same for RAGGED bags in CSR form (bag b = ids[offsets[b]:offsets[b+1]]): what the reference computes for a
    batch whose rows are padded to the longest bag with an all-zero padding row — restated bag by bag (small cases)."""
    out = []
    for b in range(offsets.numel() - 1):
        e = F.embedding(ids[int(offsets[b]):int(offsets[b + 1])].long(), table)[None]  # [1, len, D]
        out.append(seq_pool(e, mode)[0] if e.shape[1] else torch.zeros(table.shape[1], dtype=table.dtype))
    return torch.stack(out)


def _tables(sd: Dict[str, Tensor], prefix: str, enc_dict) -> Dict[str, Tensor]:
    return {c: sd[f"{prefix}{c}.weight"] for c in sparse_fields(enc_dict)}


# ------------------------------------------------------------------------------------------------
# interaction blocks  (rec_pangu/models/layers/interaction.py)
# ------------------------------------------------------------------------------------------------
def fm_bi_interaction(emb: Tensor) -> Tensor:
    """interaction.py:38-42: 0.5 * ((sum_f v)^2 - sum_f v^2) -> [B, D]."""
    s = emb.sum(dim=1)
    return (s * s - (emb * emb).sum(dim=1)) * 0.5


def fm_second_order(emb: Tensor) -> Tensor:
    """interaction.py:38-44 (product_sum_pooling) via FM_Layer interaction.py:231-235 -> [B,1]."""
    return fm_bi_interaction(emb).sum(dim=-1, keepdim=True)


def cross_net(x0: Tensor, weights: Sequence[Tensor], biases: Sequence[Tensor]) -> Tensor:
    """interaction.py:119-141: X_{l+1} = X_l + (X_l . w_l) * X_0 + b_l ; w_l given as [1,d] (Linear weight)."""
    xi = x0
    for w, b in zip(weights, biases):
        xi = xi + (xi @ w.reshape(-1, 1)) * x0 + b
    return xi


def cin(emb: Tensor, conv_w: Sequence[Tensor], conv_b: Sequence[Tensor], fc_w: Tensor, fc_b: Tensor) -> Tensor:
    """interaction.py:157-171: X_k[b,o,:] = sum_{h,m} W_k[o, h*M+m] X_0[b,h,:] X_{k-1}[b,m,:] + bias_k[o];
    no activation, no split-half; pooled over the embedding axis, concatenated, then fc -> [B,1].
    conv_w[k] is the Conv1d weight [O, H*M, 1]."""
    B, H, D = emb.shape
    x0, xi, pooled = emb, emb, []
    for w, b in zip(conv_w, conv_b):
        M = xi.shape[1]
        had = (x0.unsqueeze(2) * xi.unsqueeze(1)).reshape(B, H * M, D)  # channel c = h*M + m
        xi = torch.einsum("oc,bcd->bod", w.reshape(w.shape[0], H * M), had) + b.view(1, -1, 1)
        pooled.append(xi.sum(dim=-1))
    return torch.cat(pooled, dim=-1) @ fc_w.t() + fc_b


# ------------------------------------------------------------------------------------------------
# attention  (rec_pangu/models/layers/attention.py:35-101), AutoInt flavour: align_to="output"
# ------------------------------------------------------------------------------------------------
def mhsa(x: Tensor, wq: Tensor, wk: Tensor, wv: Tensor, wres: Optional[Tensor], num_heads: int,
         attention_dim: int, use_scale: bool = False) -> Tensor:
    """attention.py:63-95.  Quirks kept: heads are split by a RAW .view(B*H, -1, a) of the [B,T,H*a]
    projection (:73-75), no 1/sqrt(d) unless use_scale, softmax over the last axis, residual
    W_res(x) only if input_dim != H*a, and ReLU always applied at the end (:94)."""
    Bsz = x.shape[0]
    q, k, v = x @ wq.t(), x @ wk.t(), x @ wv.t()
    q = q.reshape(Bsz * num_heads, -1, attention_dim)
    k = k.reshape(Bsz * num_heads, -1, attention_dim)
    v = v.reshape(Bsz * num_heads, -1, attention_dim)
    s = torch.bmm(q, k.transpose(1, 2))
    if use_scale:
        s = s / (attention_dim ** 0.5)
    o = torch.bmm(torch.softmax(s, dim=2), v).reshape(Bsz, -1, num_heads * attention_dim)
    res = x if wres is None else x @ wres.t()
    return (o + res).relu()


# ------------------------------------------------------------------------------------------------
# MLP / LR  (rec_pangu/models/layers/deep.py:61-84, shallow.py:14-27)
# ------------------------------------------------------------------------------------------------
def mlp_relu(x: Tensor, sd: Dict[str, Tensor], prefix: str, linear_ids: Sequence[int]) -> Tensor:
    """deep.py:61-72 in eval mode with ReLU hidden activations: Linear->ReLU for all but the last id,
    plain Linear for the last (the output_dim layer). `linear_ids` are the nn.Sequential indices of
    the Linear modules ({0,2,4,6} without dropout, {0,3,6,9} with)."""
    for j, i in enumerate(linear_ids):
        x = x @ sd[f"{prefix}{i}.weight"].t() + sd[f"{prefix}{i}.bias"]
        if j + 1 < len(linear_ids):
            x = x.relu()
    return x


def dice(x: Tensor, alpha: Tensor, running_mean: Tensor, running_var: Tensor, training: bool, eps: float = 1e-9,
         momentum: float = 0.01):
    """layers/activation.py:18-34: p = sigmoid(BatchNorm1d(x, affine=False, eps=1e-9, momentum=0.01)),
    out = p x + (1 - p) alpha x.  Training: batch mean / biased variance normalise, the running statistics move by
    `momentum` (unbiased variance), as nn.BatchNorm1d does.  -> (out, running_mean', running_var')"""
    if training:
        mean, var = x.mean(dim=0), x.var(dim=0, unbiased=False)
        n = x.shape[0]
        new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
        new_rv = (1 - momentum) * running_var + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var, new_rm, new_rv = running_mean, running_var, running_mean, running_var
    p = torch.sigmoid((x - mean) / torch.sqrt(var + eps))
    return p * x + (1 - p) * alpha * x, new_rm, new_rv


def lr_layer(sd: Dict[str, Tensor], prefix: str, enc_dict, data) -> Tensor:
    """shallow.py:22-27: dim-1 embedding per field -> [B,F]; cat dense; Linear(F+ND, 1)."""
    t = _tables(sd, prefix + "emb_layer.embedding_layer.", enc_dict)
    sparse = embedding_all(t, enc_dict, data).squeeze(-1)
    z = torch.cat([sparse, get_linear_input(enc_dict, data)], dim=1)
    return z @ sd[prefix + "fc.weight"].t() + sd[prefix + "fc.bias"]


def bce_mean(pred: Tensor, label: Tensor) -> Tensor:
    """torch.nn.BCELoss() as eval()'ed at deepfm.py:31: mean of -(y log p + (1-y) log(1-p)), logs clamped at -100."""
    return F.binary_cross_entropy(pred, label)


# ------------------------------------------------------------------------------------------------
# whole models
# ------------------------------------------------------------------------------------------------
_EMB = "embedding_layer.embedding_layer."


def _dnn_input(sd, enc_dict, data):
    emb = embedding_all(_tables(sd, _EMB, enc_dict), enc_dict, data)
    return emb, torch.cat([emb.flatten(start_dim=1), get_linear_input(enc_dict, data)], dim=1)


def _linear_ids(sd, prefix):
    ids = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")
                  and sd[k].dim() == 2})
    return ids


def _finish(logit, data, is_training=True):
    pred = torch.sigmoid(logit)
    out = {"pred": pred}
    if is_training:
        out["loss"] = bce_mean(pred.squeeze(-1), data["label"])
    return out


def deepfm(sd, enc_dict, data, is_training=True):
    """ranking/deepfm.py:52-66: logit = FM2(emb) + MLP(cat(emb_flat, dense)); NO first-order term."""
    emb, x = _dnn_input(sd, enc_dict, data)
    logit = fm_second_order(emb) + mlp_relu(x, sd, "dnn.net.", _linear_ids(sd, "dnn.net."))
    return _finish(logit, data, is_training)


def fm(sd, enc_dict, data, is_training=True):
    """ranking/fm.py: logit = FM2(emb)."""
    emb = embedding_all(_tables(sd, _EMB, enc_dict), enc_dict, data)
    return _finish(fm_second_order(emb), data, is_training)


def wdl(sd, enc_dict, data, is_training=True):
    """ranking/wdl.py:57-66: logit = LR_Layer(data) + MLP(cat(emb_flat, dense))."""
    _, x = _dnn_input(sd, enc_dict, data)
    logit = lr_layer(sd, "lr.", enc_dict, data) + mlp_relu(x, sd, "dnn.net.", _linear_ids(sd, "dnn.net."))
    return _finish(logit, data, is_training)


def nfm(sd, enc_dict, data, is_training=True):
    """ranking/nfm.py:55-68: logit = LR_Layer(data) + MLP(Bi_interaction_pooling(emb)) — MLP input is [B, D]."""
    emb = embedding_all(_tables(sd, _EMB, enc_dict), enc_dict, data)
    logit = lr_layer(sd, "lr.", enc_dict, data) + mlp_relu(fm_bi_interaction(emb), sd, "dnn.net.",
                                                           _linear_ids(sd, "dnn.net."))
    return _finish(logit, data, is_training)


def dcn(sd, enc_dict, data, is_training=True):
    """ranking/dcn.py:57-67: sigmoid(fc(CrossNet(cat(emb_flat, dense)))) — no deep branch."""
    _, x = _dnn_input(sd, enc_dict, data)
    n = len([k for k in sd if k.startswith("crossnet.cross_net.") and k.endswith(".bias")])
    ws = [sd[f"crossnet.cross_net.{i}.weight.weight"] for i in range(n)]
    bs = [sd[f"crossnet.cross_net.{i}.bias"] for i in range(n)]
    c = cross_net(x, ws, bs)
    return _finish(c @ sd["fc.weight"].t() + sd["fc.bias"], data, is_training)


def xdeepfm(sd, enc_dict, data, is_training=True):
    """ranking/xdeepfm.py:60-76 (eval mode): logit = LR + CIN + MLP."""
    emb, x = _dnn_input(sd, enc_dict, data)
    n = len([k for k in sd if k.startswith("cin.cin_layer.layer_") and k.endswith(".bias")])
    cw = [sd[f"cin.cin_layer.layer_{i + 1}.weight"] for i in range(n)]
    cb = [sd[f"cin.cin_layer.layer_{i + 1}.bias"] for i in range(n)]
    logit = lr_layer(sd, "lr_layer.", enc_dict, data) + cin(emb, cw, cb, sd["cin.fc.weight"], sd["cin.fc.bias"]) \
        + mlp_relu(x, sd, "dnn.net.", _linear_ids(sd, "dnn.net."))
    return _finish(logit, data, is_training)


def autoint(sd, enc_dict, data, num_heads, attention_dim, is_training=True):
    """ranking/autoint.py:71-88 (eval mode): logit = fc(flatten(attn stack)) + MLP + LR."""
    emb, x = _dnn_input(sd, enc_dict, data)
    a, i = emb, 0
    while f"self_attention.{i}.W_q.weight" in sd:
        p = f"self_attention.{i}."
        a = mhsa(a, sd[p + "W_q.weight"], sd[p + "W_k.weight"], sd[p + "W_v.weight"],
                 sd.get(p + "W_res.weight"), num_heads, attention_dim)
        i += 1
    logit = a.flatten(start_dim=1) @ sd["fc.weight"].t() + sd["fc.bias"]
    logit = logit + mlp_relu(x, sd, "dnn.net.", _linear_ids(sd, "dnn.net."))
    logit = logit + lr_layer(sd, "lr_layer.", enc_dict, data)
    return _finish(logit, data, is_training)


def task_towers(sd, inputs, data, num_task, training, is_training, p_eps=0.0, bn_eps=1e-5):
    """The tower stack every multi-task model builds (mmoe.py:44-56 etc.): [Linear -> BatchNorm1d -> Dropout]* ->
    Linear -> Sigmoid with NO activation in between; loss = sum_t (1/T) BCE(p_t + p_eps, y_t).  `training` selects
    BatchNorm batch statistics (dropout must be 0 for a deterministic comparison)."""
    out, task_outputs = {}, []
    for t in range(num_task):
        x = inputs[t]
        p, j = f"task_{t + 1}_dnn.", 0
        while f"{p}ctr_hidden_{j}.weight" in sd:
            x = x @ sd[f"{p}ctr_hidden_{j}.weight"].t() + sd[f"{p}ctr_hidden_{j}.bias"]
            bw, bb = sd[f"{p}ctr_batchnorm_{j}.weight"], sd[f"{p}ctr_batchnorm_{j}.bias"]
            if training:
                mean, var = x.mean(dim=0), x.var(dim=0, unbiased=False)
            else:
                mean, var = sd[f"{p}ctr_batchnorm_{j}.running_mean"], sd[f"{p}ctr_batchnorm_{j}.running_var"]
            x = (x - mean) / torch.sqrt(var + bn_eps) * bw + bb
            j += 1
        x = torch.sigmoid(x @ sd[p + "task_last_layer.weight"].t() + sd[p + "task_last_layer.bias"])
        out[f"task{t + 1}_pred"] = x
        task_outputs.append(x)
    if is_training:
        loss = 0
        for t, x in enumerate(task_outputs):
            loss = loss + (1.0 / num_task) * F.binary_cross_entropy(x.squeeze(-1) + p_eps, data[f"task{t + 1}_label"])
        out["loss"] = loss
    return out


def mmoe(sd, gates, gates_bias, enc_dict, data, num_task, training=False, is_training=True):
    """multi_task/mmoe.py:81-130.  experts einsum 'ij,jkl->ikl' + bias; per-task softmax gate (gates are
    the UNREGISTERED N(0,1) tensors of mmoe.py:43-47, passed separately); gate-weighted sum over experts;
    towers; loss with p + 1e-6."""
    _, hidden = _dnn_input(sd, enc_dict, data)
    eo = torch.einsum("ij,jkl->ikl", hidden, sd["experts"]) + sd["experts_bias"]
    xs = []
    for t in range(num_task):
        g = torch.softmax(hidden @ gates[t] + gates_bias[t], dim=-1)
        xs.append((eo * g.unsqueeze(1)).sum(dim=2))
    return task_towers(sd, xs, data, num_task, training, is_training, p_eps=1e-6)


def omoe(sd, enc_dict, data, num_task, training=False, is_training=True):
    """multi_task/omoe.py:71-95: ONE input-independent gate softmax(gate[E,1], dim=0) shared by all tasks."""
    _, hidden = _dnn_input(sd, enc_dict, data)
    eo = torch.einsum("ij,jkl->ikl", hidden, sd["experts"]) + sd["experts_bias"]
    x = (eo @ torch.softmax(sd["gate"], dim=0)).squeeze(-1)
    return task_towers(sd, [x] * num_task, data, num_task, training, is_training)


def mlmmoe(sd, level_gates, gates, gates_bias, enc_dict, data, num_task, training=False, is_training=True):
    """multi_task/mlmmoe.py:88-130: level_out[..., j] = experts_out . softmax(level_gates[j], dim=0), then the MMOE
    per-task gating over the level outputs (all three gate lists are unregistered tensors, passed separately)."""
    _, hidden = _dnn_input(sd, enc_dict, data)
    eo = torch.einsum("ij,jkl->ikl", hidden, sd["experts"]) + sd["experts_bias"]
    lv = torch.cat([eo @ torch.softmax(lg, dim=0) for lg in level_gates], dim=-1)
    xs = []
    for t in range(num_task):
        g = torch.softmax(hidden @ gates[t] + gates_bias[t], dim=-1)
        xs.append((lv * g.unsqueeze(1)).sum(dim=2))
    return task_towers(sd, xs, data, num_task, training, is_training)


def sharebottom(sd, enc_dict, data, num_task, training=False, is_training=True):
    """multi_task/sharebottom.py:64-85: every tower reads cat(emb_flat, dense) directly."""
    _, hidden = _dnn_input(sd, enc_dict, data)
    return task_towers(sd, [hidden] * num_task, data, num_task, training, is_training)


# ------------------------------------------------------------------------------------------------
# optimiser  (rec_pangu/trainer.py:75 -> torch.optim.Adam, betas (.9,.999), eps 1e-8, wd 0; DENSE over all rows)
# ------------------------------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """One dense Adam update in the operation order of torch.optim.Adam's single-tensor path:
    m += (g-m)(1-b1); v = b2 v + (1-b2) g^2; p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    m = m + (g - m) * (1 - b1)
    v = v * b2 + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ------------------------------------------------------------------------------------------------
# dataset encode  (rec_pangu/dataset/base_dataset.py:47-103)
# ------------------------------------------------------------------------------------------------
def encode_sparse(values, mapping: dict):
    """base_dataset.py:92: category -> id, unseen -> vocab_size (the OOV row)."""
    oov = mapping["vocab_size"]
    return [mapping.get(x, oov) for x in values]


def encode_dense(values, lo, hi):
    """base_dataset.py:79-80: (x - min) / (max - min + 1e-5)."""
    import numpy as np
    return (np.asarray(values, dtype=np.float64) - lo) / (hi - lo + 1e-5)
