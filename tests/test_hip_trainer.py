"""RankTrainer / BenchmarkTrainer end to end on the HIP device against the fixtures captured from a run of the
reference (tests/golden/trainer.{json,npz}): same dataloaders, same seed, `device=cuda` — the training loop then
runs the HIP kernels (gather, GEMMs, loss, exact lazy Adam) and must land on the reference's metrics, final
weights, checkpoints and predictions."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden, require_gpu
from test_trainer_dataset import _frames, _loaders_in_reference_order

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


def test_rank_trainer_fit_on_hip_matches_reference_run(tmp_path):
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.trainer import RankTrainer
    meta, train_loader, valid_loader, test_loader, enc, test_df = _loaders_in_reference_order()
    ref = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    g = load_golden("trainer.npz")
    torch.manual_seed(ref["seed"])
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    n0 = hip.launch_count()
    valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=ref["epoch"], lr=ref["lr"], device=DEV)
    assert hip.launch_count() > n0 + 100, "the training loop did not run on the HIP kernels"
    assert sorted(os.listdir(tmp_path)) == ref["ckpt_files"]
    for k, v in ref["valid_metric"].items():  # 4-decimal metrics: allow the last digit to differ across devices
        assert abs(valid_metric[k] - v) <= 2e-4, (k, valid_metric[k], v)
    sd = model.state_dict()
    for k, v in g["final"].items():
        tol = 2e-4 * max(1e-2, float(v.abs().max()))
        assert (sd[k].cpu() - v).abs().max() <= tol, k
    test_metric = trainer.evaluate_model(model, test_loader, device=DEV)
    for k, v in ref["test_metric"].items():
        assert abs(test_metric[k] - v) <= 2e-4, (k, test_metric[k], v)
    with pytest.raises(RuntimeError, match="same device"):  # like the reference: batches must be moved to the device
        trainer.predict_dataloader(model, test_loader)
    p_df = trainer.predict_dataframe(model, test_df, enc, meta["schema"], device=DEV, batch_size=16)
    p_dl = trainer.predict_dataloader(model, test_loader, device=DEV)
    np.testing.assert_allclose(np.asarray(p_df), g["pred_dataframe"].numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(np.asarray(p_dl), g["pred_dataloader"].numpy(), rtol=1e-3, atol=1e-5)
    # checkpoint written from the HIP model (lazy Adam flushed by the state_dict hook) loads into a CPU model
    trainer.save_all(model, enc, str(tmp_path))
    saved = torch.load(os.path.join(tmp_path, "model.pth"), weights_only=False)
    assert sorted(saved.keys()) == ref["save_all_keys"]
    reloaded = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=saved["enc_dict"])
    reloaded.load_state_dict(saved["model"])
    np.testing.assert_allclose(np.asarray(trainer.predict_dataloader(reloaded, test_loader)), np.asarray(p_dl),
                               rtol=1e-4, atol=1e-6)


def test_multitask_trainer_on_hip(tmp_path):
    """RankTrainer(num_task=2) over every multi-task model on the HIP device: metric keys, finite losses."""
    from rec_pangu_amd.benchmark_trainer import BenchmarkTrainer
    from rec_pangu_amd.dataset import get_dataloader
    import pandas as pd
    meta, train_df, valid_df, test_df = _frames()
    schema = dict(meta["schema"], label_col=["click", "scroll"], task_type="multitask")
    train_loader, valid_loader, test_loader, enc = get_dataloader(train_df, valid_df, test_df, schema, batch_size=50)
    csv = os.path.join(tmp_path, "mt.csv")
    names = ["MMOE", "OMOE", "MLMMOE", "ShareBottom"]
    bt = BenchmarkTrainer(num_task=2, model_list=names, benchmark_res_path=csv, ckpt_root=os.path.join(tmp_path, "ck"))
    bt.run(train_loader, enc, valid_loader, test_loader, epoch=1, lr=1e-3, device=DEV)
    res = pd.read_csv(csv)
    assert list(res["model_name"]) == names
    assert {"test_task1_roc_auc_score", "test_task2_log_loss"} <= set(res.columns)
    assert np.isfinite(res["test_task1_log_loss"]).all()


def test_device_batch_loader_on_hip_through_rank_trainer(tmp_path):
    """the encoded columns live on the HIP device and every batch is a dict of device slices.  The
    shuffled loader consumes the global RNG like torch's DataLoader + RandomSampler, so RankTrainer.fit fed by it sees
    the reference run's batches in the reference run's order and must land on its metrics and final weights — with
    batches that never leave the device."""
    from rec_pangu_amd.dataset import DeviceBatchLoader
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.trainer import RankTrainer
    meta, train_loader, valid_loader, test_loader, enc, test_df = _loaders_in_reference_order()
    ref = json.load(open(os.path.join(GOLDEN, "trainer.json")))
    g = load_golden("trainer.npz")
    bs = train_loader.batch_size
    dl_train = DeviceBatchLoader(train_loader.dataset, bs, shuffle=True, device=DEV)
    dl_valid = DeviceBatchLoader(valid_loader.dataset, valid_loader.batch_size, shuffle=False, device=DEV)
    dl_test = DeviceBatchLoader(test_loader.dataset, test_loader.batch_size, shuffle=False, device=DEV)
    torch.manual_seed(5)
    first = next(iter(dl_train))
    assert all(v.is_cuda for v in first.values())
    torch.manual_seed(5)
    host = next(iter(train_loader))
    for k in host:
        assert torch.equal(first[k].cpu(), host[k]), f"batch column {k} differs from the DataLoader's"
    torch.manual_seed(ref["seed"])
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    valid_metric = trainer.fit(model, dl_train, dl_valid, epoch=ref["epoch"], lr=ref["lr"], device=DEV)
    for k, v in ref["valid_metric"].items():
        assert abs(valid_metric[k] - v) <= 2e-4, (k, valid_metric[k], v)
    sd = model.state_dict()
    for k, v in g["final"].items():
        tol = 2e-4 * max(1e-2, float(v.abs().max()))
        assert (sd[k].cpu() - v).abs().max() <= tol, k
    test_metric = trainer.evaluate_model(model, dl_test, device=DEV)
    for k, v in ref["test_metric"].items():
        assert abs(test_metric[k] - v) <= 2e-4, (k, test_metric[k], v)
    p_dl = trainer.predict_dataloader(model, dl_test, device=DEV)
    np.testing.assert_allclose(np.asarray(p_dl), g["pred_dataloader"].numpy(), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("shuffle", [False, True])
def test_pinned_batch_loader_feeds_the_same_batches(shuffle, tmp_path):
    """The host-fed path (columns in pinned host memory, double-buffered asynchronous copies on a side stream): the
    same batch dicts as torch's DataLoader in the same order, on the device, also when the consumer is slower or
    faster than the copies; and a RankTrainer.fit through it equals the fit through the device-resident loader."""
    from rec_pangu_amd.dataset import DeviceBatchLoader, PinnedBatchLoader
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.trainer import RankTrainer
    meta, train_loader, valid_loader, test_loader, enc, test_df = _loaders_in_reference_order()
    ds = train_loader.dataset
    ref_loader = torch.utils.data.DataLoader(ds, batch_size=24, shuffle=shuffle, num_workers=0)
    pinned = PinnedBatchLoader(ds, 24, shuffle=shuffle, device=DEV, depth=2)
    assert len(pinned) == len(ref_loader)
    for epoch in range(2):
        torch.manual_seed(11 + epoch)
        want = [b for b in ref_loader]
        torch.manual_seed(11 + epoch)
        got = []
        for i, b in enumerate(pinned):
            assert all(v.is_cuda for v in b.values())
            got.append({k: v.clone() for k, v in b.items()})  # (feature columns are views of recycled buffers)
            if i % 2:
                torch.cuda._sleep(2_000_000)  # a slow consumer: the next copies must not overwrite what it reads
        assert len(got) == len(want)
        for x, y in zip(want, got):
            assert list(x) == list(y)
            for k in x:
                assert x[k].dtype == y[k].dtype and torch.equal(x[k], y[k].cpu()), k
    finals = []
    for cls in (DeviceBatchLoader, PinnedBatchLoader):
        torch.manual_seed(3)
        model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
        tl = cls(ds, 32, shuffle=True, device=DEV)
        vl = cls(valid_loader.dataset, 32, shuffle=False, device=DEV)
        m = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path)).fit(model, tl, vl, epoch=2, lr=1e-2, device=DEV)
        finals.append((m, {k: v.clone() for k, v in model.state_dict().items()}))
    assert finals[0][0] == finals[1][0]
    for k in finals[0][1]:
        assert torch.equal(finals[0][1][k], finals[1][1][k]), k


@pytest.mark.gpu
def test_sort_ahead_is_bit_identical():
    require_gpu()
    """BaseModel.prefetch (the next batch's row sort on the side stream, beside the step in flight) changes no result:
    ten lazy-Adam steps with every next batch announced == the same ten steps without, bit for bit."""
    import copy
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    enc = {f"C{i}": {"vocab_size": v} for i, v in enumerate([5, 40, 3000, 17, 90000])}
    enc.update({f"I{i}": {"min": 0.0, "max": 1.0} for i in range(3)})
    torch.manual_seed(3)
    ref = DeepFM(enc_dict=enc, embedding_dim=64, hidden_units=[64, 64])
    other = copy.deepcopy(ref)
    g = torch.Generator().manual_seed(5)

    def batch():
        b = {c: torch.randint(0, enc[c]["vocab_size"] + 1, (2048,), generator=g) for c in enc if c.startswith("C")}
        b.update({c: torch.rand(2048, generator=g) for c in enc if c.startswith("I")})
        b["label"] = (torch.rand(2048, generator=g) < 0.3).float()
        return b
    host = [batch() for _ in range(10)]
    outs = []
    for model, ahead in ((ref, False), (other, True)):
        model = model.to(dev)
        opt = FusedAdam(model.parameters(), lr=1e-2)
        data = [{k: v.to(dev) for k, v in b.items()} for b in host]
        losses = []
        for i, b in enumerate(data):
            if ahead and i + 1 < len(data):
                model.prefetch(data[i + 1])
            out = model(b)
            out["loss"].backward()
            opt.step()
            model.zero_grad()
            losses.append(out["loss"].detach().clone())
        outs.append((torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_rank_trainer_fit_on_hip_on_the_reference_sample_data(tmp_path):
    """the reference's own 100-row example data, 16 + 9 column schema, 2 epochs of DeepFM(emb 16) —
    tests/golden/sample_run.* (make_golden_r3.py) — through RankTrainer.fit on the HIP path."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.trainer import RankTrainer
    from test_trainer_dataset import sample_run_loaders
    meta, g, train_loader, valid_loader, test_loader, enc = sample_run_loaders()
    torch.manual_seed(meta["seed"])
    model = DeepFM(embedding_dim=meta["embedding_dim"], enc_dict=enc)
    trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path))
    n0 = hip.launch_count()
    valid_metric = trainer.fit(model, train_loader, valid_loader, epoch=meta["epoch"], lr=meta["lr"], device=DEV)
    assert hip.launch_count() > n0 + 20
    for k, v in meta["valid_metric"].items():
        assert abs(valid_metric[k] - v) <= 2e-4, (k, valid_metric[k], v)
    sd = model.state_dict()
    for k, v in g["final"].items():
        tol = 2e-4 * max(1e-2, float(v.abs().max()))
        assert (sd[k].cpu() - v).abs().max() <= tol, k
    test_metric = trainer.evaluate_model(model, test_loader, device=DEV)
    for k, v in meta["test_metric"].items():
        assert abs(test_metric[k] - v) <= 2e-4, (k, test_metric[k], v)
    np.testing.assert_allclose(np.asarray(trainer.predict_dataloader(model, test_loader, device=DEV)),
                               g["pred_dataloader"].numpy(), rtol=1e-3, atol=1e-5)


def test_rank_trainer_fit_with_hip_graph_matches_the_eager_fit(tmp_path):
    """RankTrainer.fit(use_hip_graph=True): 3 epochs over 10 full batches + a smaller last one (which runs eagerly, as
    does the batch before it), evaluation and checkpoints between the epochs — weights and metrics equal the eager fit
    bit for bit."""
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.trainer import RankTrainer
    from rec_pangu_amd.dataset import get_dataloader
    from rec_pangu_amd.models.layers.embedding import EmbeddingLayer
    import pandas as pd
    rng = np.random.RandomState(0)
    n = 10 * 64 + 23
    df = pd.DataFrame({"a": rng.randint(0, 500, n).astype(str), "b": rng.randint(0, 7, n).astype(str),
                       "c": rng.randint(0, 3000, n).astype(str), "x": rng.rand(n), "y": rng.rand(n) * 5,
                       "click": (rng.rand(n) < 0.3).astype(int)})
    schema = {"sparse_cols": ["a", "b", "c"], "dense_cols": ["x", "y"], "label_col": "click", "task_type": "ranking"}
    finals = {}
    try:
        for use_graph in (False, True):
            torch.manual_seed(0)
            train_loader, valid_loader, _, enc = get_dataloader(df, df[:100].copy(), df[:50].copy(), schema, batch_size=64)
            torch.manual_seed(1)
            model = DeepFM(embedding_dim=16, hidden_units=[16, 8], enc_dict=enc)
            trainer = RankTrainer(num_task=1, model_ckpt_dir=str(tmp_path / f"g{int(use_graph)}"))
            metric = trainer.fit(model, train_loader, valid_loader, epoch=3, lr=2e-3, device=DEV, lr_scheduler_type="StepLR",
                                 scheduler_params={"step_size": 1, "gamma": 0.5}, use_hip_graph=use_graph)
            finals[use_graph] = (metric, {k: v.clone() for k, v in model.state_dict().items()})
    finally:
        EmbeddingLayer.unpin_sorts()
    assert finals[False][0] == finals[True][0]
    for k, v in finals[False][1].items():
        assert torch.equal(v, finals[True][1][k]), k
