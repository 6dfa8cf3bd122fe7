"""Host logic on the CPU (BASELINE config 0): dataset encode, RankTrainer.fit / evaluate / predict,
checkpoint layout and BenchmarkTrainer CSV schema against the fixtures captured from a run of the
reference (tests/golden/{dataset,trainer,benchmark}.*)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import GOLDEN, load_golden
from rec_pangu_amd.benchmark_trainer import BenchmarkTrainer
from rec_pangu_amd.dataset import BaseDataset, MultiTaskDataset, get_dataloader
from rec_pangu_amd.models.ranking import DeepFM
from rec_pangu_amd.trainer import RankTrainer

torch.set_num_threads(1)


def _frames():
    meta = json.load(open(os.path.join(GOLDEN, "dataset.json")))
    df = pd.read_json(os.path.join(GOLDEN, "dataset_frame.json"), orient="split")
    return meta, df[:100].copy(), df[100:130].copy(), df[130:].copy()


def test_dataset_encode_matches_reference():
    meta, train_df, valid_df, test_df = _frames()
    g = load_golden("dataset.npz")
    train_loader, valid_loader, test_loader, enc_dict = get_dataloader(train_df, valid_df, test_df, meta["schema"],
                                                                       batch_size=32)
    # enc_dict content equals the reference's (key ORDER is schema order here, hash order there: B4)
    assert set(enc_dict) == set(meta["enc_dict"])
    for col, ref in meta["enc_dict"].items():
        got = {str(k): (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in enc_dict[col].items()}
        assert got == ref, col
    for split, loader in (("train", train_loader), ("valid", valid_loader), ("test", test_loader)):
        for col, ref in g[split].items():
            got = loader.dataset.data_dict[col]
            assert got.dtype == ref.dtype
            if ref.dtype == torch.int64:
                assert torch.equal(got, ref), f"{split}/{col}: ids must be bit-exact"
            else:
                torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-7)
    b0 = next(iter(valid_loader))
    assert set(b0) == set(g["valid_batch0"])
    for k, ref in g["valid_batch0"].items():
        assert b0[k].dtype == ref.dtype and b0[k].shape == ref.shape
        torch.testing.assert_close(b0[k], ref, rtol=1e-6, atol=1e-7)


def _loaders_in_reference_order():
    meta, train_df, valid_df, test_df = _frames()
    train_loader, valid_loader, test_loader, enc = get_dataloader(train_df, valid_df, test_df, meta["schema"],
                                                                  batch_size=32)
    enc_ref_order = {k: enc[k] for k in meta["enc_order"]}  # the field order the reference run had
    return meta, train_loader, valid_loader, test_loader, enc_ref_order, test_df


def test_rank_trainer_fit_matches_reference_run(tmp_path):
    meta, train_loader, valid_loader, test_loader, enc, test_df = _loaders_in_reference_order()
    ref = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    g = load_golden("trainer.npz")
    torch.manual_seed(ref["seed"])
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
    for k, v in g["init"].items():
        assert torch.equal(model.state_dict()[k], v), k
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=ref["epoch"], lr=ref["lr"],
                               device=torch.device("cpu"))
    assert sorted(os.listdir(tmp_path)) == ref["ckpt_files"]
    assert valid_metric == ref["valid_metric"]
    for k, v in g["final"].items():
        torch.testing.assert_close(model.state_dict()[k], v, rtol=1e-4, atol=1e-6)
    assert trainer.evaluate_model(model, test_loader, device=torch.device("cpu")) == ref["test_metric"]
    p_df = trainer.predict_dataframe(model, test_df, enc, meta["schema"], batch_size=16)
    p_dl = trainer.predict_dataloader(model, test_loader)
    np.testing.assert_allclose(np.asarray(p_df), g["pred_dataframe"].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(np.asarray(p_dl), g["pred_dataloader"].numpy(), rtol=1e-4, atol=1e-6)
    trainer.save_all(model, enc, str(tmp_path))
    saved = torch.load(os.path.join(tmp_path, "model.pth"), weights_only=False)
    assert sorted(saved.keys()) == ref["save_all_keys"]
    reloaded = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=saved["enc_dict"])
    reloaded.load_state_dict(saved["model"])
    assert trainer.predict_dataloader(reloaded, test_loader) == p_dl


def test_early_stopping_and_scheduler(tmp_path):
    _, train_loader, valid_loader, _, enc, _ = _loaders_in_reference_order()
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=4, hidden_units=[8], enc_dict=enc)
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    m = trainer.fit(model, train_loader, valid_loader, epoch=6, lr=1e-2, use_earlystopping=True, max_patience=1,
                    monitor_metric="roc_auc_score", lr_scheduler_type="StepLR",
                    scheduler_params={"step_size": 1, "gamma": 0.5})
    assert set(m) == {"roc_auc_score", "log_loss"}
    assert "model_best.pth" in os.listdir(tmp_path)
    with pytest.raises(ValueError):
        trainer.fit(model, train_loader, valid_loader, epoch=1, lr_scheduler_type="nope")
    with pytest.raises(AssertionError):
        trainer.fit(model, train_loader, valid_loader, epoch=1, use_earlystopping=True, monitor_metric="f1")


def test_benchmark_trainer_csv_schema(tmp_path):
    _, train_loader, valid_loader, test_loader, enc, _ = _loaders_in_reference_order()
    ref = json.load(open(os.path.join(GOLDEN, "benchmark.json")))
    csv = os.path.join(tmp_path, "bench.csv")
    torch.manual_seed(1)
    bt = BenchmarkTrainer(num_task=1, model_list=["DeepFM", "FM"], benchmark_res_path=csv,
                          ckpt_root=os.path.join(tmp_path, "ck"))
    bt.run(train_loader, enc, valid_loader, test_loader, epoch=1, lr=1e-3, device=torch.device("cpu"))
    res = pd.read_csv(csv)
    assert list(res.columns) == ref["columns"]
    assert list(res["model_name"]) == ref["model_name"]
    assert sorted(os.listdir(os.path.join(tmp_path, "ck"))) == ref["ckpt_dirs"]
    assert sorted(os.listdir(os.path.join(tmp_path, "ck", "DeepFM"))) == ref["ckpt_files"]
    with pytest.raises(NameError):
        BenchmarkTrainer(model_list=["NoSuchModel"], benchmark_res_path=csv).run(train_loader, enc)


def test_multitask_dataset_and_trainer(tmp_path):
    """The reference's MultiTaskDataset is dead at v0.4.1 (B1); ours implements its documented output."""
    from rec_pangu_amd.models.multi_task import MMOE
    meta, train_df, valid_df, test_df = _frames()
    schema = dict(meta["schema"], label_col=["click", "scroll"], task_type="multitask")
    train_loader, valid_loader, test_loader, enc = get_dataloader(train_df, valid_df, test_df, schema, batch_size=50)
    b = next(iter(valid_loader))
    assert {"task1_label", "task2_label"} <= set(b) and "label" not in b
    assert isinstance(train_loader.dataset, MultiTaskDataset)
    torch.manual_seed(0)
    model = MMOE(num_task=2, n_expert=2, embedding_dim=4, mmoe_hidden_dim=8, hidden_dim=[8, 4], enc_dict=enc)
    trainer = RankTrainer(num_task=2, model_ckpt_dir=str(tmp_path))
    m = trainer.fit(model, train_loader, valid_loader, epoch=1, lr=1e-3)
    assert set(m) == {"test_task1_roc_auc_score", "test_task1_log_loss", "test_task2_roc_auc_score",
                      "test_task2_log_loss"}
    preds = trainer.predict_dataloader(model, test_loader)
    assert len(preds) == 2 and len(preds[0]) == len(test_df)


def test_benchmark_trainer_multitask_list(tmp_path):
    """BenchmarkTrainer(num_task=2) over the multi-task models, ShareBottom included (its
    constructor takes no `device`, which the reference passes anyway — B2)."""
    meta, train_df, valid_df, test_df = _frames()
    schema = dict(meta["schema"], label_col=["click", "scroll"], task_type="multitask")
    train_loader, valid_loader, test_loader, enc = get_dataloader(train_df, valid_df, test_df, schema, batch_size=50)
    csv = os.path.join(tmp_path, "mt.csv")
    names = ["MMOE", "OMOE", "MLMMOE", "ShareBottom"]
    bt = BenchmarkTrainer(num_task=2, model_list=names, benchmark_res_path=csv, ckpt_root=os.path.join(tmp_path, "ck"))
    bt.run(train_loader, enc, valid_loader, test_loader, epoch=1, lr=1e-3, device=torch.device("cpu"))
    res = pd.read_csv(csv)
    assert list(res["model_name"]) == names
    assert {"test_task1_roc_auc_score", "test_task2_log_loss", "train_model_time"} <= set(res.columns)


def test_device_batch_loader_yields_the_dataloader_batches(tmp_path):
    """DeviceBatchLoader = the same batch dicts as torch's DataLoader over the same dataset,
    without the per-sample __getitem__/collate; RankTrainer takes it in place of a DataLoader."""
    from rec_pangu_amd.dataset import DeviceBatchLoader
    meta, train_loader, valid_loader, test_loader, enc, _ = _loaders_in_reference_order()
    fast = DeviceBatchLoader(valid_loader.dataset, batch_size=valid_loader.batch_size, shuffle=False)
    assert len(fast) == len(valid_loader)
    for a, b in zip(fast, valid_loader):
        assert set(a) == set(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    shuffled = DeviceBatchLoader(train_loader.dataset, batch_size=32, shuffle=True,
                                 generator=torch.Generator().manual_seed(0))
    seen = torch.cat([b["label"] for b in shuffled])
    assert seen.numel() == len(train_loader.dataset)
    assert torch.equal(seen.sort().values, train_loader.dataset._label.sort().values)
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=4, hidden_units=[8], enc_dict=enc)
    m = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path)).fit(model, shuffled, fast, epoch=1, lr=1e-3)
    assert set(m) == {"roc_auc_score", "log_loss"}


def test_device_batch_loader_reproduces_the_reference_run(tmp_path):
    """With shuffle=True and no explicit generator DeviceBatchLoader consumes the global RNG exactly like torch's
    DataLoader + RandomSampler (one base seed per iter(), one sampler seed + randperm per shuffled epoch), so a
    RankTrainer.fit fed by it reproduces the reference's captured 2-epoch run: same batches in the same order, same
    metrics, same final weights."""
    from rec_pangu_amd.dataset import DeviceBatchLoader
    meta, train_loader, valid_loader, test_loader, enc, _ = _loaders_in_reference_order()
    torch.manual_seed(7)
    a = [b for b in train_loader] + [b for b in valid_loader] + [b for b in train_loader]
    torch.manual_seed(7)
    ft = DeviceBatchLoader(train_loader.dataset, train_loader.batch_size, shuffle=True)
    fv = DeviceBatchLoader(valid_loader.dataset, valid_loader.batch_size, shuffle=False)
    b = [x for x in ft] + [x for x in fv] + [x for x in ft]
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for k in x:
            assert torch.equal(x[k], y[k]), k
    ref = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    g = load_golden("trainer.npz")
    torch.manual_seed(ref["seed"])
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
    valid_metric = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path)).fit(
        model, ft, fv, epoch=ref["epoch"], lr=ref["lr"], device=torch.device("cpu"))
    assert valid_metric == ref["valid_metric"]
    for k, v in g["final"].items():
        torch.testing.assert_close(model.state_dict()[k], v, rtol=1e-4, atol=1e-6)


def test_baseline_config1_shape_on_cpu(tmp_path):
    """BASELINE.json configs[0] as stated: DeepFM on a synthetic CSV-like frame with 13 dense + 26 sparse columns,
    vocabularies <= 1e4, emb_dim = 16, bsz = 512, through get_dataloader -> RankTrainer.fit on the CPU (plumbing, no
    GPU).  Checks the shapes that configuration implies, the first training batch against the CPU oracle on the same
    weights, and that two epochs run through the reference's loop (metrics keys, checkpoints, a falling loss)."""
    from oracle import ref_ops as R
    rng = np.random.default_rng(0)
    n = 3000
    card = [int(c) for c in rng.integers(3, 10000, size=26)]
    frame = {f"I{i + 1}": rng.gamma(2.0, 3.0, size=n).astype(np.float32) for i in range(13)}
    for j, c in enumerate(card):
        # Zipf-ish ids as strings (like Criteo's hashed categories); ids 0..c-1, the tail rarely seen
        frame[f"C{j + 1}"] = np.char.add("v", np.minimum(rng.zipf(1.3, size=n) - 1, c - 1).astype(str))
    z = 0.4 * (frame["I1"] - 6) / 4 + (np.char.equal(frame["C1"], "v0")).astype(np.float32) - 0.8
    frame["label"] = (rng.random(n) < 1 / (1 + np.exp(-z))).astype(np.float32)
    df = pd.DataFrame(frame)
    schema = {"sparse_cols": [f"C{j + 1}" for j in range(26)], "dense_cols": [f"I{i + 1}" for i in range(13)],
              "label_col": "label", "task_type": "ranking"}
    train_loader, valid_loader, test_loader, enc = get_dataloader(df[:2048].copy(), df[2048:2560].copy(),
                                                                  df[2560:].copy(), schema, batch_size=512)
    assert sum("vocab_size" in v for v in enc.values()) == 26 and sum("min" in v for v in enc.values()) == 13
    assert max(v["vocab_size"] for v in enc.values() if "vocab_size" in v) <= 10000
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=16, hidden_units=[64, 64, 64], enc_dict=enc)
    assert model.dnn_input_dim == 26 * 16 + 13
    torch.manual_seed(1)
    b0 = next(iter(train_loader))
    assert b0["C1"].shape == (512,) and b0["C1"].dtype == torch.int64 and b0["I1"].dtype == torch.float32
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ref = R.deepfm(sd, enc, b0)
    out = model(b0)
    torch.testing.assert_close(out["pred"], ref["pred"], rtol=0, atol=1e-6)
    torch.testing.assert_close(out["loss"], ref["loss"], rtol=0, atol=1e-6)
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    first = trainer.evaluate_model(model, valid_loader)
    metric = trainer.fit(model, train_loader, valid_loader, epoch=2, lr=1e-3, device=torch.device("cpu"))
    assert set(metric) == {"roc_auc_score", "log_loss"}
    assert sorted(os.listdir(tmp_path)) == ["model_e_1.pth", "model_e_2.pth"]
    assert metric["log_loss"] < first["log_loss"], (first, metric)
    preds = trainer.predict_dataloader(model, test_loader)
    assert len(preds) == len(df) - 2560 and all(0.0 <= p <= 1.0 for p in preds)


def test_one_ahead_feeds_the_same_batches_and_announces_the_next():
    """model_pipeline._one_ahead (the look-ahead of train_model): the loader's batches in order, each one yielded only after
    the NEXT one has been announced to the model (BaseModel.prefetch); loaders that recycle buffers are told to keep two
    batches alive; a model without `prefetch` (stock torch modules) and an empty loader are fine."""
    from rec_pangu_amd.model_pipeline import _one_ahead

    class Loader(list):
        hold = 1

    class Model:
        def __init__(self):
            self.log = []

        def prefetch(self, batch):
            self.log.append(("announce", int(batch["i"])))

    batches = Loader({"i": torch.tensor(i)} for i in range(4))
    model = Model()
    seen = []
    for b in _one_ahead(batches, torch.device("cpu"), model):
        model.log.append(("step", int(b["i"])))
        seen.append(int(b["i"]))
    assert seen == [0, 1, 2, 3]
    assert batches.hold == 2
    assert model.log == [("announce", 1), ("step", 0), ("announce", 2), ("step", 1), ("announce", 3), ("step", 2), ("step", 3)]
    assert [int(b["i"]) for b in _one_ahead([{"i": torch.tensor(7)}], torch.device("cpu"), object())] == [7]
    assert list(_one_ahead([], torch.device("cpu"), model)) == []
    # a CPU-resident model ignores the announcement (BASELINE config 0: plumbing, no GPU)
    enc = {"C1": {"vocab_size": 5}, "I1": {"min": 0.0, "max": 1.0}}
    cpu_model = DeepFM(enc_dict=enc, embedding_dim=4, hidden_units=[8])
    cpu_model.prefetch({"C1": torch.zeros(3, dtype=torch.long), "I1": torch.zeros(3)})
    assert getattr(cpu_model.embedding_layer, "_ahead", None) is None


# ---- SURVEY a18's own fixture: the reference's 100-row example data with the 16 + 9 column schema of
# examples/ranking/run_ranking_example.py:17-24 (tests/golden/sample_run.*, written by make_golden_r3.py) ----------------
def sample_run_fixture():
    meta = json.load(open(os.path.join(GOLDEN, "sample_run.json")))
    fr = meta["frame"]
    df = pd.DataFrame(fr["data"], columns=fr["columns"], index=fr["index"])
    a, b, c = meta["splits"]
    return meta, df[:a], df[:b], df[:c], load_golden("sample_run.npz")


def sample_run_loaders():
    meta, train_df, valid_df, test_df, g = sample_run_fixture()
    loaders = get_dataloader(train_df, valid_df, test_df, meta["schema"], batch_size=meta["batch_size"])
    enc = {k: loaders[3][k] for k in meta["enc_order"]}  # the field order the reference run had (B4)
    return meta, g, loaders[0], loaders[1], loaders[2], enc


def test_sample_data_encode_matches_reference():
    meta, train_df, valid_df, test_df, g = sample_run_fixture()
    train_loader, valid_loader, test_loader, enc_dict = get_dataloader(train_df, valid_df, test_df, meta["schema"],
                                                                       batch_size=meta["batch_size"])
    assert set(enc_dict) == set(meta["enc_dict"])
    for col, ref in meta["enc_dict"].items():
        got = {str(k): (int(v) if isinstance(v, (int, np.integer)) else float(v)) for k, v in enc_dict[col].items()}
        assert got == ref, col
    for split, loader in (("train", train_loader), ("valid", valid_loader), ("test", test_loader)):
        cols = {k.split("/", 1)[1]: v for k, v in g["enc"].items() if k.startswith(split + "/")}
        assert set(cols) == set(loader.dataset.data_dict)
        for col, ref in cols.items():
            got = loader.dataset.data_dict[col]
            assert got.dtype == ref.dtype, (split, col)
            if ref.dtype == torch.int64:
                assert torch.equal(got, ref), f"{split}/{col}: ids must be bit-exact"
            else:
                torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-7)


def test_rank_trainer_fit_on_the_reference_sample_data(tmp_path):
    meta, g, train_loader, valid_loader, test_loader, enc = sample_run_loaders()
    torch.manual_seed(meta["seed"])
    model = DeepFM(embedding_dim=meta["embedding_dim"], enc_dict=enc)
    for k, v in g["init"].items():
        assert torch.equal(model.state_dict()[k], v), k
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=meta["epoch"], lr=meta["lr"],
                               device=torch.device("cpu"))
    assert valid_metric == meta["valid_metric"]
    for k, v in g["final"].items():
        torch.testing.assert_close(model.state_dict()[k], v, rtol=1e-4, atol=1e-6)
    assert trainer.evaluate_model(model, test_loader, device=torch.device("cpu")) == meta["test_metric"]
    np.testing.assert_allclose(np.asarray(trainer.predict_dataloader(model, test_loader)), g["pred_dataloader"].numpy(),
                               rtol=1e-4, atol=1e-6)


def test_one_ahead_pairs_for_the_graphed_loop():
    """model_pipeline._one_ahead(pairs=True): (batch, next batch) pairs for a GraphedTrainStep — the same batches in the
    same order, the last one paired with None, nothing announced to the model (the captured step sorts the next batch)."""
    from rec_pangu_amd.model_pipeline import _one_ahead

    class _Model:
        def __init__(self):
            self.announced = 0

        def prefetch(self, data):
            self.announced += 1

    loader = [{"a": torch.tensor([i])} for i in range(5)]
    m = _Model()
    pairs = list(_one_ahead(loader, torch.device("cpu"), m, pairs=True))
    assert [int(c["a"]) for c, _ in pairs] == [0, 1, 2, 3, 4]
    assert [None if n is None else int(n["a"]) for _, n in pairs] == [1, 2, 3, 4, None]
    assert all(pairs[i][1] is pairs[i + 1][0] for i in range(4)), "the announced batch IS the next current batch (identity)"
    assert m.announced == 0
    assert list(_one_ahead([], torch.device("cpu"), m, pairs=True)) == []
