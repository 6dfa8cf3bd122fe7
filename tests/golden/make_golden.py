#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING the upstream reference.

Run in the build container only (needs /root/reference):

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

It imports the reference's own classes (rec_pangu.models.*, rec_pangu.trainer, rec_pangu.dataset),
feeds them small seeded inputs and stores inputs + outputs as plain arrays (npz) and json.
Only data is written: no reference source, bytecode or pickled reference objects.

What each file pins (reference file:line of the code that produced it):
  model_<name>.npz   rec_pangu/models/ranking/{deepfm,xdeepfm,dcn,autoint,fm}.py forward,
                     rec_pangu/models/multi_task/mmoe.py forward/loss, autograd grads of `loss`,
                     and 2 steps of torch.optim.Adam as built at rec_pangu/trainer.py:75
  layers.npz         rec_pangu/models/layers/{embedding,interaction,attention,deep,shallow}.py
  dataset.npz/.json  rec_pangu/dataset/base_dataset.py:47-103 on a synthetic DataFrame
  trainer.json/.npz  rec_pangu/trainer.py:51-122 + rec_pangu/model_pipeline.py:17-219 (2 epochs)
  benchmark.json     rec_pangu/benchmark_trainer.py:86-95 csv columns
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
from rec_pangu.models.ranking import DeepFM, xDeepFM, DCN, AutoInt, FM, WDL, NFM  # noqa: E402
from rec_pangu.models.multi_task import MMOE, OMOE, MLMMOE, ShareBottom  # noqa: E402
from rec_pangu.models.layers import (EmbeddingLayer, InnerProductLayer, CrossNet,  # noqa: E402
                                     CompressedInteractionNet, MultiHeadSelfAttention, MLP, LR_Layer)
from rec_pangu.trainer import RankTrainer  # noqa: E402
from rec_pangu.benchmark_trainer import BenchmarkTrainer  # noqa: E402
from rec_pangu.dataset import get_dataloader  # noqa: E402

torch.set_num_threads(1)

# ----------------------------------------------------------------------------------------------
# shared small schema: dense and sparse keys deliberately interleaved to pin enc_dict-order use
# ----------------------------------------------------------------------------------------------
ENC_ORDER = ["I1", "C1", "C2", "I2", "C3", "I3", "C4", "C5"]
VOCAB = {"C1": 7, "C2": 3, "C3": 50, "C4": 11, "C5": 2}
B = 24


def small_enc_dict():
    enc = {}
    for k in ENC_ORDER:
        if k.startswith("I"):
            enc[k] = {"min": 0.0, "max": 1.0}
        else:
            enc[k] = {"vocab_size": VOCAB[k]}
    return enc


def small_batch(seed=7):
    g = torch.Generator().manual_seed(seed)
    data = {}
    for k in ENC_ORDER:
        if k.startswith("I"):
            data[k] = torch.rand(B, generator=g)
        else:
            # inclusive upper bound: id == vocab_size is the OOV row (base_dataset.py:92)
            data[k] = torch.randint(0, VOCAB[k] + 1, (B,), generator=g)
    data["label"] = (torch.rand(B, generator=g) < 0.3).float()
    data["task1_label"] = (torch.rand(B, generator=g) < 0.3).float()
    data["task2_label"] = (torch.rand(B, generator=g) < 0.5).float()
    return data


def to_np(t):
    return t.detach().cpu().numpy().copy()


ONLY = None  # set from the command line: regenerate just these model cases


def dump_model_case(name, build, seed=1234, train_mode=False, extra=None):
    if ONLY is not None and name not in ONLY:
        return
    out = {}
    torch.manual_seed(seed)
    model = build()
    for k, v in model.state_dict().items():
        out["init/" + k] = to_np(v)
    if extra is not None:
        extra(model, out)
    data = small_batch()
    for k, v in data.items():
        out["batch/" + k] = to_np(v)
    model.train(train_mode)
    res = model({k: v.clone() for k, v in data.items()})
    for k, v in res.items():
        out["out/" + k] = to_np(v)
    model.zero_grad()
    res["loss"].backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = to_np(p.grad)
    # state after the forward/backward above (BatchNorm running stats move in train mode)
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["after1/" + k] = to_np(v)
    # two optimiser steps exactly as RankTrainer.fit builds the optimiser (trainer.py:75)
    torch.manual_seed(seed)
    model = build()
    model.train(train_mode)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
    for _ in range(2):
        r = model({k: v.clone() for k, v in data.items()})
        r["loss"].backward()
        opt.step()
        model.zero_grad()
    for k, v in model.state_dict().items():
        out["adam2/" + k] = to_np(v)
    # inference path (is_training=False) after the two steps
    model.eval()
    with torch.no_grad():
        r = model({k: v.clone() for k, v in data.items()}, is_training=False)
    for k, v in r.items():
        out["adam2_out/" + k] = to_np(v)
    np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)
    print("wrote", name, len(out), "arrays")


def mmoe_extra(model, out):
    # B3: gates are plain python lists, not in state_dict -> dump separately
    for i, g in enumerate(model.gates):
        out[f"gates/{i}"] = to_np(g)
    for i, g in enumerate(model.gates_bias):
        out[f"gates_bias/{i}"] = to_np(g)


def mlmmoe_extra(model, out):
    mmoe_extra(model, out)
    for i, g in enumerate(model.level_gates):
        out[f"level_gates/{i}"] = to_np(g)


def make_models():
    enc = small_enc_dict()
    dump_model_case("deepfm", lambda: DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), train_mode=True)
    dump_model_case("fm", lambda: FM(embedding_dim=8, enc_dict=enc), train_mode=True)
    dump_model_case("wdl", lambda: WDL(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), train_mode=True)
    dump_model_case("nfm", lambda: NFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), train_mode=True)
    dump_model_case("dcn", lambda: DCN(embedding_dim=8, crossing_layers=3, enc_dict=enc), train_mode=True)
    # default MLP dropout 0.1 in xDeepFM/AutoInt -> compare in eval()
    dump_model_case("xdeepfm", lambda: xDeepFM(embedding_dim=8, dnn_hidden_units=[16, 8],
                                               cin_layer_units=[6, 4], enc_dict=enc), train_mode=False)
    dump_model_case("autoint_h2", lambda: AutoInt(embedding_dim=8, dnn_hidden_units=[16, 8], attention_layers=2,
                                                  num_heads=2, attention_dim=4, enc_dict=enc), train_mode=False)
    dump_model_case("autoint_h1", lambda: AutoInt(embedding_dim=8, dnn_hidden_units=[16, 8], attention_layers=1,
                                                  num_heads=1, attention_dim=8, enc_dict=enc), train_mode=False)
    dump_model_case("autoint_h3a5", lambda: AutoInt(embedding_dim=8, dnn_hidden_units=[16], attention_layers=2,
                                                    num_heads=3, attention_dim=5, enc_dict=enc), train_mode=False)
    dump_model_case("mmoe_eval", lambda: MMOE(num_task=2, n_expert=3, embedding_dim=8, mmoe_hidden_dim=16,
                                              hidden_dim=[8, 4], dropouts=[0.2, 0.2], enc_dict=enc,
                                              device=torch.device("cpu")), train_mode=False, extra=mmoe_extra)
    # train-mode BatchNorm (batch statistics) pinned with dropout disabled
    dump_model_case("mmoe_train", lambda: MMOE(num_task=2, n_expert=4, embedding_dim=8, mmoe_hidden_dim=16,
                                               hidden_dim=[8, 4], dropouts=[0.0, 0.0], enc_dict=enc,
                                               device=torch.device("cpu")), train_mode=True, extra=mmoe_extra)


    # the sibling multi-task models on the same expert-GEMM / tower kernels
    for tag, tm, dp in (("eval", False, [0.2, 0.2]), ("train", True, [0.0, 0.0])):
        dump_model_case(f"omoe_{tag}", lambda: OMOE(num_task=2, n_expert=3, embedding_dim=8, omoe_hidden_dim=16,
                                                    hidden_dim=[8, 4], dropouts=dp, enc_dict=enc,
                                                    device=torch.device("cpu")), train_mode=tm)
        dump_model_case(f"mlmmoe_{tag}", lambda: MLMMOE(num_task=2, n_expert=3, embedding_dim=8, mmoe_hidden_dim=16,
                                                        hidden_dim=[8, 4], dropouts=dp, enc_dict=enc,
                                                        device=torch.device("cpu")), train_mode=tm,
                        extra=mlmmoe_extra)
        dump_model_case(f"sharebottom_{tag}", lambda: ShareBottom(num_task=2, embedding_dim=8, hidden_units=[8, 4],
                                                                  dropouts=dp, enc_dict=enc), train_mode=tm)


def make_layers():
    out = {}
    enc = small_enc_dict()
    data = small_batch(seed=11)
    g = torch.Generator().manual_seed(3)
    for k, v in data.items():
        out["batch/" + k] = to_np(v)

    torch.manual_seed(5)
    emb = EmbeddingLayer(enc, 8)
    for k, v in emb.state_dict().items():
        out["emb/w/" + k] = to_np(v)
    out["emb/all"] = to_np(emb(data))
    out["emb/by_name_C3"] = to_np(emb(data, name="C3"))
    seq = torch.randint(0, VOCAB["C3"] + 1, (B, 4), generator=g)
    out["emb/seq_in"] = to_np(seq)
    d2 = dict(data)
    d2["C3_seq"] = seq
    out["emb/by_name_C3_seq"] = to_np(emb(d2, name="C3_seq"))

    x = torch.randn(B, 5, 8, generator=g)
    out["ip/in"] = to_np(x)
    out["ip/product_sum_pooling"] = to_np(InnerProductLayer(output="product_sum_pooling")(x))
    out["ip/Bi_interaction_pooling"] = to_np(InnerProductLayer(output="Bi_interaction_pooling")(x))

    torch.manual_seed(6)
    cn = CrossNet(43, 3)
    with torch.no_grad():
        for p in cn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    x0 = torch.randn(B, 43, generator=g, requires_grad=True)
    y = cn(x0)
    (y * torch.linspace(0.5, 1.5, 43)).sum().backward()
    out["cross/in"] = to_np(x0)
    out["cross/out"] = to_np(y)
    out["cross/grad_in"] = to_np(x0.grad)
    for k, p in cn.named_parameters():
        out["cross/w/" + k] = to_np(p)
        out["cross/gw/" + k] = to_np(p.grad)

    torch.manual_seed(7)
    cin = CompressedInteractionNet(5, [6, 4], output_dim=1)
    xe = torch.randn(B, 5, 8, generator=g, requires_grad=True)
    yc = cin(xe)
    (yc.squeeze(-1) * torch.linspace(-1, 1, B)).sum().backward()
    out["cin/in"] = to_np(xe)
    out["cin/out"] = to_np(yc)
    out["cin/grad_in"] = to_np(xe.grad)
    for k, p in cin.named_parameters():
        out["cin/w/" + k] = to_np(p)
        out["cin/gw/" + k] = to_np(p.grad)

    for tag, (din, heads, adim) in {"a": (8, 2, 4), "b": (8, 2, 3), "c": (8, 1, 8), "d": (6, 3, 5)}.items():
        torch.manual_seed(8)
        att = MultiHeadSelfAttention(din, attention_dim=adim, num_heads=heads, align_to="output")
        xa = torch.randn(B, 5, din, generator=g, requires_grad=True)
        ya = att(xa)
        (ya * ya).sum().backward()
        out[f"attn_{tag}/in"] = to_np(xa)
        out[f"attn_{tag}/out"] = to_np(ya)
        out[f"attn_{tag}/grad_in"] = to_np(xa.grad)
        for k, p in att.named_parameters():
            out[f"attn_{tag}/w/" + k] = to_np(p)
            out[f"attn_{tag}/gw/" + k] = to_np(p.grad)

    torch.manual_seed(9)
    mlp = MLP(input_dim=43, output_dim=1, hidden_units=[16, 8], hidden_activations="relu", dropout_rates=0)
    xm = torch.randn(B, 43, generator=g, requires_grad=True)
    ym = mlp(xm)
    (ym.squeeze(-1) * torch.linspace(-1, 1, B)).sum().backward()
    out["mlp/in"] = to_np(xm)
    out["mlp/out"] = to_np(ym)
    out["mlp/grad_in"] = to_np(xm.grad)
    for k, p in mlp.named_parameters():
        out["mlp/w/" + k] = to_np(p)
        out["mlp/gw/" + k] = to_np(p.grad)
    torch.manual_seed(9)
    mlp2 = MLP(input_dim=43, output_dim=None, hidden_units=[16, 8], hidden_activations=["tanh", "sigmoid"],
               dropout_rates=0.1, batch_norm=True, output_activation=None)
    mlp2.eval()
    out["mlp2/out"] = to_np(mlp2(xm.detach()))
    for k, v in mlp2.state_dict().items():
        out["mlp2/w/" + k] = to_np(v)

    torch.manual_seed(10)
    lr = LR_Layer(enc)
    for k, v in lr.state_dict().items():
        out["lr/w/" + k] = to_np(v)
    ylr = lr(data)
    ylr.sum().backward()
    out["lr/out"] = to_np(ylr)
    for k, p in lr.named_parameters():
        out["lr/gw/" + k] = to_np(p.grad)

    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)
    print("wrote layers", len(out), "arrays")


# ----------------------------------------------------------------------------------------------
# dataset / trainer fixtures on a synthetic DataFrame (our own generator, not a reference file)
# ----------------------------------------------------------------------------------------------
def synth_frame(n=160, seed=0):
    rng = np.random.RandomState(seed)
    df = pd.DataFrame({
        "user": ["u%d" % i for i in rng.randint(0, 23, n)],
        "item": ["i%d" % i for i in rng.randint(0, 9, n)],
        "city": rng.randint(100, 105, n),
        "hour": rng.randint(0, 24, n),
        "expo": np.round(rng.gamma(2.0, 10.0, n), 2),
        "clk": np.round(rng.gamma(1.0, 3.0, n), 2),
        "dur": rng.randint(1, 300, n),
    })
    z = 0.03 * df["clk"] - 0.01 * df["expo"] + 0.3 * (df["city"] == 101) + rng.randn(n) * 0.5
    df["click"] = (z > np.median(z)).astype(int)
    df["scroll"] = (rng.rand(n) < 0.4).astype(int)
    return df


SCHEMA = {"sparse_cols": ["user", "item", "city", "hour"], "dense_cols": ["expo", "clk", "dur"],
          "label_col": "click", "task_type": "ranking"}


def jsonable_enc(enc):
    o = {}
    for k, v in enc.items():
        o[k] = {str(kk): (int(vv) if isinstance(vv, (int, np.integer)) else float(vv)) for kk, vv in v.items()}
    return o


def make_dataset_and_trainer():
    df = synth_frame()
    train_df, valid_df, test_df = df[:100].copy(), df[100:130].copy(), df[130:].copy()
    torch.manual_seed(0)
    train_loader, valid_loader, test_loader, enc_dict = get_dataloader(train_df, valid_df, test_df, SCHEMA,
                                                                       batch_size=32)
    meta = {"enc_order": list(enc_dict.keys()), "enc_dict": jsonable_enc(enc_dict), "schema": SCHEMA,
            "pythonhashseed": os.environ.get("PYTHONHASHSEED")}
    arrs = {}
    for split, loader in (("train", train_loader), ("valid", valid_loader), ("test", test_loader)):
        for col, t in loader.dataset.data_dict.items():
            arrs[f"{split}/{col}"] = to_np(t)
    # one collated, unshuffled batch as the DataLoader delivers it
    b0 = next(iter(valid_loader))
    for k, v in b0.items():
        arrs["valid_batch0/" + k] = to_np(v)
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **arrs)
    df.to_json(os.path.join(HERE, "dataset_frame.json"), orient="split")
    with open(os.path.join(HERE, "dataset.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=False)

    # RankTrainer.fit, 2 epochs, DeepFM(emb 8): pins the epoch loop, Adam hyper-params, metric keys/rounding
    tarr = {}
    torch.manual_seed(42)
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc_dict)
    for k, v in model.state_dict().items():
        tarr["init/" + k] = to_np(v)
    with tempfile.TemporaryDirectory() as td:
        trainer = RankTrainer(num_task=1, model_ckpt_dir=td)
        valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=2, lr=1e-2,
                                   device=torch.device("cpu"))
        ckpt_files = sorted(os.listdir(td))
        test_metric = trainer.evaluate_model(model, test_loader, device=torch.device("cpu"))
        preds_df = trainer.predict_dataframe(model, test_df, enc_dict, SCHEMA, batch_size=16)
        preds_dl = trainer.predict_dataloader(model, test_loader)
        trainer.save_all(model, enc_dict, td)
        saved = torch.load(os.path.join(td, "model.pth"), weights_only=False)
        saved_keys = sorted(saved.keys())
    for k, v in model.state_dict().items():
        tarr["final/" + k] = to_np(v)
    tarr["pred_dataframe"] = np.asarray(preds_df, dtype=np.float32)
    tarr["pred_dataloader"] = np.asarray(preds_dl, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "trainer.npz"), **tarr)
    with open(os.path.join(HERE, "trainer.json"), "w") as f:
        json.dump({"valid_metric": valid_metric, "test_metric": test_metric, "ckpt_files": ckpt_files,
                   "save_all_keys": saved_keys, "seed": 42, "epoch": 2, "lr": 1e-2, "batch_size": 32}, f, indent=1)
    print("trainer:", valid_metric, test_metric, ckpt_files)

    # BenchmarkTrainer csv schema (DataFrame.append was removed in pandas 2 -> shim it for the run)
    if not hasattr(pd.DataFrame, "append"):
        def _append(self, other, ignore_index=False):
            return pd.concat([self, pd.DataFrame([other])], ignore_index=ignore_index)
        pd.DataFrame.append = _append
    with tempfile.TemporaryDirectory() as td:
        csv = os.path.join(td, "bench.csv")
        torch.manual_seed(1)
        bt = BenchmarkTrainer(num_task=1, model_list=["DeepFM", "FM"], benchmark_res_path=csv,
                              ckpt_root=os.path.join(td, "ck"))
        bt.run(train_loader, enc_dict, valid_loader, test_loader, epoch=1, lr=1e-3, device=torch.device("cpu"))
        res = pd.read_csv(csv)
        ck = sorted(os.listdir(os.path.join(td, "ck")))
        ck_inner = sorted(os.listdir(os.path.join(td, "ck", "DeepFM")))
    with open(os.path.join(HERE, "benchmark.json"), "w") as f:
        json.dump({"columns": list(res.columns), "model_name": list(res["model_name"]), "ckpt_dirs": ck,
                   "ckpt_files": ck_inner}, f, indent=1)
    print("benchmark columns:", list(res.columns))


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `make_golden.py wdl nfm`: only those model_<name>.npz files
        ONLY = set(sys.argv[1:])
        make_models()
    else:
        make_models()
        make_layers()
        make_dataset_and_trainer()
