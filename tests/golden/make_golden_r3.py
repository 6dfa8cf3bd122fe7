#!/usr/bin/env python
"""Round-3 golden vectors, produced by RUNNING the upstream reference (build container only):

    PYTHONHASHSEED=0 python tests/golden/make_golden_r3.py

  pool.npz            rec_pangu/models/layers/embedding.py:64-71 (`_seq` lookup -> [B, L, D]) followed by
                      rec_pangu/models/layers/sequence.py:13-36 (MaskedAveragePooling) and :38-59 (MaskedSumPooling),
                      forward and the autograd gradient of the table — the pin of rp_embed_gather_pool_fwd / _bwd
  sample_run.json/.npz  SURVEY a18: RankTrainer.fit (rec_pangu/trainer.py:51-122, model_pipeline.py:17-219), 2 epochs of
                      DeepFM(embedding_dim=16) on the reference's own 100-row example data
                      (examples/ranking/sample_data/ranking_sample_data.csv) with the 16 + 9 column schema of
                      examples/ranking/run_ranking_example.py:17-24 and its 80 / 90 / 95-row splits, seed 0.
                      Stored: the raw columns of the schema (data), the encoded arrays the reference's dataset
                      produced, the enc_dict and its key order, initial and final weights, metrics, predictions.
Only data is written: no reference source, bytecode or pickled reference objects.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()

import pandas as pd  # noqa: E402
import torch  # noqa: E402
from rec_pangu.models.ranking import DeepFM  # noqa: E402
from rec_pangu.models.layers import EmbeddingLayer  # noqa: E402
from rec_pangu.models.layers.sequence import MaskedAveragePooling, MaskedSumPooling  # noqa: E402
from rec_pangu.trainer import RankTrainer  # noqa: E402
from rec_pangu.dataset import get_dataloader  # noqa: E402

torch.set_num_threads(1)


def to_np(t):
    return t.detach().cpu().numpy().copy()


def make_pool():
    out = {}
    g = torch.Generator().manual_seed(21)
    enc = {"C1": {"vocab_size": 7}, "I1": {"min": 0.0, "max": 1.0}, "hist": {"vocab_size": 60}, "C2": {"vocab_size": 3}}
    for case, (B, L, D, zero_pad_row) in {"a": (24, 5, 8, False), "b": (37, 9, 64, True), "c": (16, 3, 6, True)}.items():
        torch.manual_seed(7)
        emb = EmbeddingLayer(enc, D)
        if zero_pad_row:  # id 0 = padding with an all-zero row: what "mask by zeros" (sequence.py:51) relies on
            with torch.no_grad():
                emb.embedding_layer["hist"].weight[0].zero_()
                emb.embedding_layer["hist"].weight[5, 1] = 0.0  # a zero ELEMENT of a real row: the mask is per element
        seq = torch.randint(1, 61, (B, L), generator=g)
        lens = torch.randint(0, L + 1, (B,), generator=g)  # some bags empty (all padding), some full
        seq = seq * (torch.arange(L)[None, :] < lens[:, None])
        for k, v in emb.state_dict().items():
            out[f"{case}/w/{k}"] = to_np(v)
        out[f"{case}/seq"] = to_np(seq)
        X = {"hist_seq": seq}
        for name, pool in (("sum", MaskedSumPooling()), ("avg", MaskedAveragePooling())):
            emb.zero_grad()
            e = emb(X, name="hist_seq")
            y = pool(e)
            cot = torch.randn(y.shape, generator=g)
            (y * cot).sum().backward()
            out[f"{case}/{name}/out"] = to_np(y)
            out[f"{case}/{name}/cot"] = to_np(cot)
            out[f"{case}/{name}/grad"] = to_np(emb.embedding_layer["hist"].weight.grad)
        out[f"{case}/lookup"] = to_np(emb(X, name="hist_seq"))
    np.savez_compressed(os.path.join(HERE, "pool.npz"), **out)
    print("wrote pool", len(out), "arrays")


SCHEMA = {
    "sparse_cols": ['user_id', 'item_id', 'item_type', 'dayofweek', 'is_workday', 'city', 'county',
                    'town', 'village', 'lbs_city', 'lbs_district', 'hardware_platform', 'hardware_ischarging',
                    'os_type', 'network_type', 'position'],
    "dense_cols": ['item_expo_1d', 'item_expo_7d', 'item_expo_14d', 'item_expo_30d', 'item_clk_1d',
                   'item_clk_7d', 'item_clk_14d', 'item_clk_30d', 'use_duration'],
    "label_col": 'click',
    'task_type': 'ranking'}


def jsonable_enc(enc):
    o = {}
    for k, v in enc.items():
        o[k] = {str(kk): (int(vv) if isinstance(vv, (int, np.integer)) else float(vv)) for kk, vv in v.items()}
    return o


def make_sample_run():
    df = pd.read_csv("/root/reference/examples/ranking/sample_data/ranking_sample_data.csv")
    cols = SCHEMA["sparse_cols"] + SCHEMA["dense_cols"] + [SCHEMA["label_col"]]
    raw = df[cols]
    train_df, valid_df, test_df = df[:80], df[:90], df[:95]  # run_ranking_example.py:26-28
    torch.manual_seed(0)
    train_loader, valid_loader, test_loader, enc_dict = get_dataloader(train_df, valid_df, test_df, SCHEMA, batch_size=512)
    arrs = {}
    for split, loader in (("train", train_loader), ("valid", valid_loader), ("test", test_loader)):
        for col, t in loader.dataset.data_dict.items():
            arrs[f"enc/{split}/{col}"] = to_np(t)
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=16, enc_dict=enc_dict)
    for k, v in model.state_dict().items():
        arrs["init/" + k] = to_np(v)
    with tempfile.TemporaryDirectory() as td:
        trainer = RankTrainer(num_task=1, model_ckpt_dir=td)
        valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=2, lr=1e-3, device=torch.device("cpu"))
        test_metric = trainer.evaluate_model(model, test_loader, device=torch.device("cpu"))
        preds = trainer.predict_dataloader(model, test_loader)
    for k, v in model.state_dict().items():
        arrs["final/" + k] = to_np(v)
    arrs["pred_dataloader"] = np.asarray(preds, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "sample_run.npz"), **arrs)
    meta = {"schema": SCHEMA, "enc_order": list(enc_dict.keys()), "enc_dict": jsonable_enc(enc_dict),
            "valid_metric": valid_metric, "test_metric": test_metric, "seed": 0, "epoch": 2, "lr": 1e-3, "batch_size": 512,
            "embedding_dim": 16, "splits": [80, 90, 95], "pythonhashseed": os.environ.get("PYTHONHASHSEED"),
            "frame": json.loads(raw.to_json(orient="split"))}
    with open(os.path.join(HERE, "sample_run.json"), "w") as f:
        json.dump(meta, f, indent=None, ensure_ascii=True)
    print("sample run:", valid_metric, test_metric)


if __name__ == "__main__":
    make_pool()
    make_sample_run()
