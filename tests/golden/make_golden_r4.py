#!/usr/bin/env python
"""Round-4 golden vectors, produced by RUNNING the upstream reference (build container only):

    PYTHONHASHSEED=0 python tests/golden/make_golden_r4.py

  dice.npz        rec_pangu/models/layers/activation.py:10-34 (Dice): forward + autograd gradients in training mode
                  (batch statistics, running statistics after the step), a forward in eval mode (running statistics),
                  and rec_pangu/models/layers/deep.py:11-84 (MLP) with Dice instances as hidden activations
  adam_long.npz   rec_pangu/trainer.py:75 (torch.optim.Adam(lr=1e-3, betas=(0.9, 0.999), eps=1e-8): DENSE Adam over every
                  table row every step) on a tiny reference DeepFM for 600 steps in the call order of
                  model_pipeline.py:52-58 (forward, backward, step, zero_grad), with batches that leave most table rows
                  untouched for hundreds of steps — the reference-generated pin of the lazy / closed-form / deferred
                  execution of the table optimizer: weights at steps 300 and 600, every batch.
Only data is written: no reference source, bytecode or pickled reference objects.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

import torch  # noqa: E402
from rec_pangu.models.layers.activation import Dice  # noqa: E402
from rec_pangu.models.layers import MLP  # noqa: E402
from rec_pangu.models.ranking import DeepFM  # noqa: E402

torch.set_num_threads(1)


def to_np(t):
    return t.detach().cpu().numpy().copy()


def make_dice():
    out = {}
    g = torch.Generator().manual_seed(5)
    for case, (M, N) in {"a": (32, 6), "b": (257, 40)}.items():
        torch.manual_seed(3)
        d = Dice(N)
        with torch.no_grad():
            d.alpha.copy_(0.5 * torch.randn(N, generator=g))  # (zeros at init: the gate would hide alpha's gradient path)
        x = (2.0 * torch.randn(M, N, generator=g) + 0.3).requires_grad_(True)
        cot = torch.randn(M, N, generator=g)
        d.train()
        y = d(x)
        (y * cot).sum().backward()
        out[f"{case}/x"], out[f"{case}/alpha"], out[f"{case}/cot"] = to_np(x), to_np(d.alpha), to_np(cot)
        out[f"{case}/train/y"], out[f"{case}/train/dx"], out[f"{case}/train/dalpha"] = to_np(y), to_np(x.grad), to_np(d.alpha.grad)
        out[f"{case}/running_mean"], out[f"{case}/running_var"] = to_np(d.bn.running_mean), to_np(d.bn.running_var)
        out[f"{case}/num_batches_tracked"] = to_np(d.bn.num_batches_tracked)
        d.eval()
        x2 = x.detach().clone().requires_grad_(True)
        d.alpha.grad = None
        y2 = d(x2)
        (y2 * cot).sum().backward()
        out[f"{case}/eval/y"], out[f"{case}/eval/dx"], out[f"{case}/eval/dalpha"] = to_np(y2), to_np(x2.grad), to_np(d.alpha.grad)
    # an MLP whose hidden activations are Dice instances (deep.py:36-47 accepts module objects)
    torch.manual_seed(11)
    mlp = MLP(input_dim=10, output_dim=1, hidden_units=[16, 8], hidden_activations=[Dice(16), Dice(8)], dropout_rates=0)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, Dice):
                m.alpha.copy_(0.3 * torch.randn(m.alpha.shape, generator=g))
    for k, v in mlp.state_dict().items():
        out["mlp/w/" + k] = to_np(v)
    x = torch.randn(48, 10, generator=g).requires_grad_(True)
    cot = torch.randn(48, 1, generator=g)
    mlp.train()
    y = mlp(x)
    (y * cot).sum().backward()
    out["mlp/x"], out["mlp/cot"], out["mlp/y"], out["mlp/dx"] = to_np(x), to_np(cot), to_np(y), to_np(x.grad)
    for k, p in mlp.named_parameters():
        out["mlp/g/" + k] = to_np(p.grad)
    for k, v in mlp.state_dict().items():
        out["mlp/after/" + k] = to_np(v)  # (running statistics moved)
    np.savez_compressed(os.path.join(HERE, "dice.npz"), **out)
    print("wrote dice", len(out), "arrays")


sys.path.insert(0, os.path.dirname(HERE))
from conftest import ADAM_LONG_ENC  # noqa: E402  (the ordered enc_dict, shared with the tests that replay the run)


def make_adam_long():
    steps, B, D = 600, 12, 8
    out = {}
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=D, hidden_units=[16, 8], enc_dict=ADAM_LONG_ENC)
    for k, v in model.state_dict().items():
        out["init/" + k] = to_np(v)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)  # trainer.py:75
    g = torch.Generator().manual_seed(17)
    cols = {c: [] for c in list(ADAM_LONG_ENC) + ["label"]}
    model.train()
    losses = []
    for t in range(1, steps + 1):
        batch = {}
        for c, e in ADAM_LONG_ENC.items():
            if "vocab_size" in e:
                batch[c] = torch.randint(0, e["vocab_size"] + 1, (B,), generator=g)
            else:
                batch[c] = torch.rand(B, generator=g)
        batch["label"] = (torch.rand(B, generator=g) < 0.3).float()
        for c in cols:
            cols[c].append(to_np(batch[c]))
        o = model(batch)  # model_pipeline.py:52
        o["loss"].backward()
        opt.step()
        model.zero_grad()
        losses.append(float(o["loss"]))
        if t in (300, 600):
            for k, v in model.state_dict().items():
                out[f"step{t}/" + k] = to_np(v)
    for c in cols:
        out["batch/" + c] = np.stack(cols[c])
    out["loss"] = np.asarray(losses, dtype=np.float32)
    # a probe batch after training: predictions of the final weights (eval: no dropout in DeepFM anyway)
    probe = {c: torch.from_numpy(out["batch/" + c][:50].reshape(-1)) for c in cols}
    with torch.no_grad():
        out["probe_pred"] = to_np(model(probe, is_training=False)["pred"])
    np.savez_compressed(os.path.join(HERE, "adam_long.npz"), **out)
    C1 = out["batch/C1"]
    last_touch = {}
    for t in range(steps):
        for r in C1[t]:
            last_touch[int(r)] = t + 1
    gaps = sorted(steps - v for v in last_touch.values())
    print("wrote adam_long", len(out), "arrays; C1 rows touched:", len(last_touch), "median steps owed at the end:", gaps[len(gaps) // 2],
          "max:", gaps[-1], "rows owed > 256:", sum(x > 256 for x in gaps))


if __name__ == "__main__":
    make_dice()
    make_adam_long()
