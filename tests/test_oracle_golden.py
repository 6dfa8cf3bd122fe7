"""The oracle (oracle/ref_ops.py) against golden vectors produced by RUNNING the reference
(tests/golden/make_golden.py).  CPU only.  Tolerances: index work bit-exact; fp32 blocks 1e-6
relative-ish (same ATen arithmetic, possibly different association); logits/loss 1e-5 abs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, small_enc_dict
from oracle import ref_ops as R

torch.set_num_threads(1)
TOL = dict(rtol=1e-5, atol=1e-6)


def _grads(loss, sd, names):
    gs = torch.autograd.grad(loss, [sd[n] for n in names], allow_unused=True)
    return dict(zip(names, gs))


def _model_case(name, fn):
    g = load_golden(f"model_{name}.npz")
    enc = small_enc_dict()
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in g["init"].items()}
    out = fn(sd, enc, g["batch"], g)
    for k, v in g["out"].items():
        torch.testing.assert_close(out[k].detach(), v, rtol=1e-5, atol=1e-6, msg=lambda m: f"{name}:{k}: {m}")
    names = [k for k in g["grad"].keys() if k in sd]
    gr = _grads(out["loss"], sd, names)
    for k in names:
        got = gr[k] if gr[k] is not None else torch.zeros_like(sd[k])
        torch.testing.assert_close(got, g["grad"][k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{name}:grad {k}: {m}")
    return g, sd


def test_deepfm():
    _model_case("deepfm", lambda sd, enc, b, g: R.deepfm(sd, enc, b))


def test_fm():
    _model_case("fm", lambda sd, enc, b, g: R.fm(sd, enc, b))


def test_wdl():
    _model_case("wdl", lambda sd, enc, b, g: R.wdl(sd, enc, b))


def test_nfm():
    _model_case("nfm", lambda sd, enc, b, g: R.nfm(sd, enc, b))


def test_dcn():
    _model_case("dcn", lambda sd, enc, b, g: R.dcn(sd, enc, b))


def test_xdeepfm():
    _model_case("xdeepfm", lambda sd, enc, b, g: R.xdeepfm(sd, enc, b))


@pytest.mark.parametrize("tag,h,a", [("autoint_h2", 2, 4), ("autoint_h1", 1, 8), ("autoint_h3a5", 3, 5)])
def test_autoint(tag, h, a):
    _model_case(tag, lambda sd, enc, b, g: R.autoint(sd, enc, b, h, a))


@pytest.mark.parametrize("tag,training", [("mmoe_eval", False), ("mmoe_train", True)])
def test_mmoe(tag, training):
    def run(sd, enc, b, g):
        gates = [g["gates"][str(i)] for i in range(2)]
        gb = [g["gates_bias"][str(i)] for i in range(2)]
        return R.mmoe(sd, gates, gb, enc, b, num_task=2, training=training)
    _model_case(tag, run)


@pytest.mark.parametrize("training", [False, True])
def test_omoe_mlmmoe_sharebottom(training):
    tag = "train" if training else "eval"
    _model_case(f"omoe_{tag}", lambda sd, enc, b, g: R.omoe(sd, enc, b, num_task=2, training=training))
    _model_case(f"sharebottom_{tag}", lambda sd, enc, b, g: R.sharebottom(sd, enc, b, num_task=2, training=training))

    def run(sd, enc, b, g):
        lst = lambda grp: [g[grp][str(i)] for i in range(len(g[grp]))]  # noqa: E731
        return R.mlmmoe(sd, lst("level_gates"), lst("gates"), lst("gates_bias"), enc, b, num_task=2,
                        training=training)
    _model_case(f"mlmmoe_{tag}", run)


def test_deepfm_two_adam_steps():
    """Dense Adam as RankTrainer builds it (trainer.py:75), two steps, against the reference's own run."""
    g = load_golden("model_deepfm.npz")
    enc = small_enc_dict()
    p = {k: v.clone() for k, v in g["init"].items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in (1, 2):
        sd = {k: t.clone().requires_grad_(True) for k, t in p.items()}
        loss = R.deepfm(sd, enc, g["batch"])["loss"]
        gr = torch.autograd.grad(loss, list(sd.values()))
        for (k, _), gk in zip(sd.items(), gr):
            p[k], m[k], v2[k] = R.adam_step(p[k], gk, m[k], v2[k], step, lr=1e-2)
    for k, ref in g["adam2"].items():
        torch.testing.assert_close(p[k], ref, rtol=1e-5, atol=1e-6, msg=lambda s: f"adam2 {k}: {s}")
    out = R.deepfm(p, enc, g["batch"], is_training=False)
    torch.testing.assert_close(out["pred"], g["adam2_out"]["pred"], rtol=1e-5, atol=1e-6)


def test_layers():
    g = load_golden("layers.npz")
    enc = small_enc_dict()
    b = g["batch"]
    tables = {k.split(".")[1]: v for k, v in g["emb"].items() if k.startswith("w/")}
    # index work: bit-exact
    assert torch.equal(R.embedding_all(tables, enc, b), g["emb"]["all"])
    assert torch.equal(R.embedding_by_name(tables, b, "C3"), g["emb"]["by_name_C3"])
    d2 = dict(b)
    d2["C3_seq"] = g["emb"]["seq_in"]
    assert torch.equal(R.embedding_by_name(tables, d2, "C3_seq"), g["emb"]["by_name_C3_seq"])
    with pytest.raises(IndexError):
        bad = dict(b)
        bad["C2"] = b["C2"].clone()
        bad["C2"][3] = 4  # vocab 3 -> rows 0..3
        R.embedding_all(tables, enc, bad)

    torch.testing.assert_close(R.fm_second_order(g["ip"]["in"]), g["ip"]["product_sum_pooling"], **TOL)
    torch.testing.assert_close(R.fm_bi_interaction(g["ip"]["in"]), g["ip"]["Bi_interaction_pooling"], **TOL)

    c = g["cross"]
    ws = [c[f"w/cross_net.{i}.weight.weight"] for i in range(3)]
    bs = [c[f"w/cross_net.{i}.bias"] for i in range(3)]
    torch.testing.assert_close(R.cross_net(c["in"], ws, bs), c["out"], rtol=1e-5, atol=1e-5)

    c = g["cin"]
    cw = [c[f"w/cin_layer.layer_{i}.weight"] for i in (1, 2)]
    cb = [c[f"w/cin_layer.layer_{i}.bias"] for i in (1, 2)]
    torch.testing.assert_close(R.cin(c["in"], cw, cb, c["w/fc.weight"], c["w/fc.bias"]), c["out"], rtol=1e-5, atol=1e-5)

    for tag, (h, a) in {"a": (2, 4), "b": (2, 3), "c": (1, 8), "d": (3, 5)}.items():
        c = g[f"attn_{tag}"]
        y = R.mhsa(c["in"], c["w/W_q.weight"], c["w/W_k.weight"], c["w/W_v.weight"], c.get("w/W_res.weight"), h, a)
        torch.testing.assert_close(y, c["out"], rtol=1e-5, atol=1e-5)

    c = g["mlp"]
    sd = {k[2:]: v for k, v in c.items() if k.startswith("w/")}
    torch.testing.assert_close(R.mlp_relu(c["in"], sd, "net.", [0, 2, 4]), c["out"], **TOL)

    c = g["lr"]
    sd = {"lr." + k[2:]: v for k, v in c.items() if k.startswith("w/")}
    torch.testing.assert_close(R.lr_layer(sd, "lr.", enc, b), c["out"], **TOL)


def test_dataset_encode():
    import json, os
    from conftest import GOLDEN
    import pandas as pd
    meta = json.load(open(os.path.join(GOLDEN, "dataset.json")))
    df = pd.read_json(os.path.join(GOLDEN, "dataset_frame.json"), orient="split")
    g = load_golden("dataset.npz")
    enc = meta["enc_dict"]
    valid = df[100:130]
    train = df[:100]
    for col in meta["schema"]["sparse_cols"]:
        ids = R.encode_sparse(train[col].astype(str).tolist(), enc[col])  # train split IS cast (base_dataset.py:58)
        assert np.array_equal(np.asarray(ids, dtype=np.int64), g["train"][col].numpy())
    for col in meta["schema"]["sparse_cols"]:
        # reference quirk (B8): with a given enc_dict the column is NOT cast to str (the cast lives in
        # get_enc_dict, base_dataset.py:58), so numeric-typed categorical columns all map to OOV here.
        ids = R.encode_sparse(valid[col].tolist(), enc[col])
        assert np.array_equal(np.asarray(ids, dtype=np.int64), g["valid"][col].numpy())
    for col in meta["schema"]["dense_cols"]:
        x = R.encode_dense(valid[col].values, enc[col]["min"], enc[col]["max"]).astype(np.float32)
        np.testing.assert_allclose(x, g["valid"][col].numpy(), rtol=1e-6, atol=1e-7)


POOL_CASES = ("a", "b", "c")


def pool_case(g, case):
    """(table [61, D], seq [B, L], {mode: (out, cot, grad)}) of one case of tests/golden/pool.npz"""
    c = g[case]
    return c["w/embedding_layer.hist.weight"], c["seq"], {m: (c[f"{m}/out"], c[f"{m}/cot"], c[f"{m}/grad"])
                                                          for m in ("sum", "avg")}


@pytest.mark.parametrize("case", POOL_CASES)
def test_seq_pooling(case):
    """the `_seq` lookup + MaskedSumPooling / MaskedAveragePooling restatement against the reference's outputs and
    autograd gradients (tests/golden/pool.npz, make_golden_r3.py)"""
    table, seq, modes = pool_case(load_golden("pool.npz"), case)
    assert torch.equal(R.embedding_by_name({"hist": table}, {"hist_seq": seq}, "hist_seq"), load_golden("pool.npz")[case]["lookup"])
    for mode, (out, cot, grad) in modes.items():
        w = table.clone().requires_grad_(True)
        y = R.embedding_seq_pooled({"hist": w}, {"hist_seq": seq}, "hist_seq", "sum" if mode == "sum" else "average")
        assert torch.equal(y, out), mode  # same ATen ops, same order
        (y * cot).sum().backward()
        torch.testing.assert_close(w.grad, grad, rtol=1e-6, atol=0)
    # ragged form: dropping the padding ids (whose row is all zero in cases b, c) gives the same sums / averages
    if case != "a":
        keep = seq != 0
        lens = keep.sum(1)
        offsets = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
        for mode, (out, _, _) in modes.items():
            y = R.embedding_bags_pooled(table, seq[keep], offsets, "sum" if mode == "sum" else "average")
            torch.testing.assert_close(y, out, rtol=1e-6, atol=1e-7)


# ---- round 4: Dice (activation.py:10-34) and the 600-step dense-Adam run of the reference ---------------------------
@pytest.mark.parametrize("case", ["a", "b"])
def test_dice_oracle_and_cpu_module_vs_reference(case):
    from rec_pangu_amd.models.layers import Dice
    g = load_golden("dice.npz")[case]
    N = g["x"].shape[1]
    for mode in ("train", "eval"):
        x = g["x"].clone().requires_grad_(True)
        alpha = g["alpha"].clone().requires_grad_(True)
        rm = torch.zeros(N) if mode == "train" else g["running_mean"]
        rv = torch.ones(N) if mode == "train" else g["running_var"]
        y, nrm, nrv = R.dice(x, alpha, rm, rv, training=(mode == "train"))
        (y * g["cot"]).sum().backward()
        torch.testing.assert_close(y.detach(), g[f"{mode}/y"], **TOL)
        torch.testing.assert_close(x.grad, g[f"{mode}/dx"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(alpha.grad, g[f"{mode}/dalpha"], rtol=1e-4, atol=1e-5)
        if mode == "train":
            torch.testing.assert_close(nrm, g["running_mean"], **TOL)
            torch.testing.assert_close(nrv, g["running_var"], **TOL)
    # the drop-in module on CPU (BASELINE config 0): same keys, same numbers
    d = Dice(N)
    assert list(d.state_dict().keys()) == ["alpha", "bn.running_mean", "bn.running_var", "bn.num_batches_tracked"]
    with torch.no_grad():
        d.alpha.copy_(g["alpha"])
    d.train()
    x = g["x"].clone().requires_grad_(True)
    y = d(x)
    (y * g["cot"]).sum().backward()
    torch.testing.assert_close(y.detach(), g["train/y"], **TOL)
    torch.testing.assert_close(d.alpha.grad, g["train/dalpha"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(d.bn.running_var, g["running_var"], **TOL)


def test_cpu_deepfm_reproduces_the_reference_600_step_adam_run():
    """tests/golden/adam_long.npz: the reference's DeepFM under the reference's dense Adam (trainer.py:75) for 600 steps.
    The CPU drop-in (BASELINE config 0: same modules, torch.optim.Adam from make_adam) must land on the same weights."""
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import make_adam
    from conftest import ADAM_LONG_ENC
    g = load_golden("adam_long.npz")
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=ADAM_LONG_ENC)
    model.load_state_dict(g["init"])
    opt = make_adam(model, 1e-3)
    assert type(opt) is torch.optim.Adam
    cols = list(g["batch"].keys())
    for t in range(1, 601):
        batch = {c: g["batch"][c][t - 1] for c in cols}
        out = model(batch)
        out["loss"].backward()
        opt.step()
        model.zero_grad()
        if t in (300, 600):
            for k, v in model.state_dict().items():
                torch.testing.assert_close(v, g[f"step{t}"][k], rtol=1e-5, atol=1e-6, msg=lambda m: f"step {t} {k}: {m}")
