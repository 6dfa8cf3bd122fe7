"""Drop-in surface on the CPU (BASELINE config 0, "plumbing, no GPU"): constructor signatures,
state_dict keys/shapes, init RNG stream and forward/backward numerics of rec_pangu_amd.models against
the golden vectors produced by running the reference (tests/golden/make_golden.py)."""
import inspect

import pytest
import torch

from conftest import load_golden, small_enc_dict
from rec_pangu_amd.models.ranking import DeepFM, xDeepFM, DCN, AutoInt, FM, WDL, NFM, LR
from rec_pangu_amd.models.multi_task import MMOE, OMOE, MLMMOE, ShareBottom

torch.set_num_threads(1)

CASES = {
    "deepfm": (lambda enc: DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), True),
    "fm": (lambda enc: FM(embedding_dim=8, enc_dict=enc), True),
    "wdl": (lambda enc: WDL(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), True),
    "nfm": (lambda enc: NFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc), True),
    "dcn": (lambda enc: DCN(embedding_dim=8, crossing_layers=3, enc_dict=enc), True),
    "xdeepfm": (lambda enc: xDeepFM(embedding_dim=8, dnn_hidden_units=[16, 8], cin_layer_units=[6, 4], enc_dict=enc), False),
    "autoint_h2": (lambda enc: AutoInt(embedding_dim=8, dnn_hidden_units=[16, 8], attention_layers=2, num_heads=2,
                                       attention_dim=4, enc_dict=enc), False),
    "autoint_h1": (lambda enc: AutoInt(embedding_dim=8, dnn_hidden_units=[16, 8], attention_layers=1, num_heads=1,
                                       attention_dim=8, enc_dict=enc), False),
    "autoint_h3a5": (lambda enc: AutoInt(embedding_dim=8, dnn_hidden_units=[16], attention_layers=2, num_heads=3,
                                         attention_dim=5, enc_dict=enc), False),
    "mmoe_eval": (lambda enc: MMOE(num_task=2, n_expert=3, embedding_dim=8, mmoe_hidden_dim=16, hidden_dim=[8, 4],
                                   dropouts=[0.2, 0.2], enc_dict=enc, device=torch.device("cpu")), False),
    "mmoe_train": (lambda enc: MMOE(num_task=2, n_expert=4, embedding_dim=8, mmoe_hidden_dim=16, hidden_dim=[8, 4],
                                    dropouts=[0.0, 0.0], enc_dict=enc, device=torch.device("cpu")), True),
}
for _tag, _tm, _dp in (("eval", False, [0.2, 0.2]), ("train", True, [0.0, 0.0])):
    CASES[f"omoe_{_tag}"] = (lambda enc, dp=_dp: OMOE(num_task=2, n_expert=3, embedding_dim=8, omoe_hidden_dim=16,
                                                      hidden_dim=[8, 4], dropouts=dp, enc_dict=enc,
                                                      device=torch.device("cpu")), _tm)
    CASES[f"mlmmoe_{_tag}"] = (lambda enc, dp=_dp: MLMMOE(num_task=2, n_expert=3, embedding_dim=8, mmoe_hidden_dim=16,
                                                          hidden_dim=[8, 4], dropouts=dp, enc_dict=enc,
                                                          device=torch.device("cpu")), _tm)
    CASES[f"sharebottom_{_tag}"] = (lambda enc, dp=_dp: ShareBottom(num_task=2, embedding_dim=8, hidden_units=[8, 4],
                                                                    dropouts=dp, enc_dict=enc), _tm)


def build(name, seed=1234):
    torch.manual_seed(seed)
    return CASES[name][0](small_enc_dict())


@pytest.mark.parametrize("name", list(CASES))
def test_init_stream_and_state_dict_contract(name):
    """Same seed -> same initial weights as the reference: same parameters, registered in the same
    order with the same shapes, drawing from the global RNG in the same sequence."""
    g = load_golden(f"model_{name}.npz")
    model = build(name)
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["init"].keys())
    for k, v in g["init"].items():
        assert sd[k].shape == v.shape, k
        assert torch.equal(sd[k], v), f"{name}: init of {k} differs from the reference's"
    if "gates" in g:  # MMOE / MLMMOE: tensors the reference keeps in plain lists (B3)
        for i in range(2):
            assert torch.equal(model.gates[i], g["gates"][str(i)])
            assert torch.equal(model.gates_bias[i], g["gates_bias"][str(i)])
        for i, lg in g.get("level_gates", {}).items():
            assert torch.equal(model.level_gates[int(i)], lg)
        assert not any(k.startswith("_gate") for k in sd)
        assert not any("_gate" in n for n, _ in model.named_parameters())


@pytest.mark.parametrize("name", list(CASES))
def test_forward_backward_adam_vs_reference(name):
    g = load_golden(f"model_{name}.npz")
    train_mode = CASES[name][1]
    model = build(name)
    model.train(train_mode)
    data = {k: v.clone() for k, v in g["batch"].items()}
    out = model(data)
    for k, v in g["out"].items():
        torch.testing.assert_close(out[k].detach(), v, rtol=1e-5, atol=1e-6, msg=lambda m: f"{name}:{k}: {m}")
    model.zero_grad()
    out["loss"].backward()
    params = dict(model.named_parameters())
    for k, v in g["grad"].items():
        got = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
        torch.testing.assert_close(got, v, rtol=1e-4, atol=1e-6, msg=lambda m: f"{name}:grad {k}: {m}")
    for k, v in g.get("after1", {}).items():
        torch.testing.assert_close(model.state_dict()[k], v, rtol=1e-5, atol=1e-6)
    # two optimiser steps as RankTrainer.fit builds the optimiser (trainer.py:75)
    model = build(name)
    model.train(train_mode)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
    for _ in range(2):
        r = model({k: v.clone() for k, v in g["batch"].items()})
        r["loss"].backward()
        opt.step()
        model.zero_grad()
    sd = model.state_dict()
    for k, v in g["adam2"].items():
        torch.testing.assert_close(sd[k], v, rtol=1e-4, atol=1e-6, msg=lambda m: f"{name}:adam2 {k}: {m}")
    model.eval()
    with torch.no_grad():
        r = model({k: v.clone() for k, v in g["batch"].items()}, is_training=False)
    assert "loss" not in r
    for k, v in g["adam2_out"].items():
        torch.testing.assert_close(r[k], v, rtol=1e-4, atol=1e-6)


def test_constructor_signatures():
    """SURVEY.md §8b, captured from the reference with inspect.signature."""
    def sig(c):
        return {k: v.default for k, v in inspect.signature(c.__init__).parameters.items() if k != "self"}
    assert sig(DeepFM) == dict(embedding_dim=32, hidden_units=[64, 64, 64], loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(xDeepFM) == dict(embedding_dim=32, dnn_hidden_units=[64, 64, 64], cin_layer_units=[16, 16, 16],
                                loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(DCN) == dict(embedding_dim=32, hidden_units=[64, 64, 64], crossing_layers=3,
                            loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(AutoInt) == dict(embedding_dim=32, dnn_hidden_units=[64, 64, 64], attention_layers=1, num_heads=1,
                                attention_dim=8, loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(FM) == dict(embedding_dim=32, loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(WDL) == sig(NFM) == sig(DeepFM)
    assert sig(LR) == dict(loss_fun='torch.nn.BCELoss()', enc_dict=None)
    assert sig(MLMMOE) == sig(MMOE)
    assert sig(OMOE) == dict(num_task=2, n_expert=3, embedding_dim=40, omoe_hidden_dim=128, expert_activation=None,
                             hidden_dim=[128, 64], dropouts=[0.2, 0.2], enc_dict=None, device=None)
    assert sig(ShareBottom) == dict(num_task=2, embedding_dim=40, hidden_units=[128, 64], dropouts=[0.2, 0.2],
                                    enc_dict=None)
    assert sig(MMOE) == dict(num_task=2, n_expert=3, embedding_dim=40, mmoe_hidden_dim=128, expert_activation=None,
                             hidden_dim=[128, 64], dropouts=[0.2, 0.2], enc_dict=None, device=None)


def test_embedding_arena_views_survive_moves_and_replacement():
    enc = small_enc_dict()
    torch.manual_seed(0)
    m = DeepFM(embedding_dim=8, hidden_units=[8], enc_dict=enc)
    layer = m.embedding_layer
    ref = {c: layer.embedding_layer[c].weight.detach().clone() for c in layer.emb_feature}
    m = m.double().float()  # goes through _apply: one arena move, views re-pointed
    off = 0
    for c in layer.emb_feature:
        w = layer.embedding_layer[c].weight
        assert w.data_ptr() == layer.arena.data_ptr() + off * 8 * 4
        assert torch.equal(w, ref[c])
        off += w.shape[0]
    # load_state_dict copies in place -> still views
    sd = {k: v + 1 for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert torch.equal(layer.arena[:8], ref["C1"] + 1)
    # set_weights with the reference's one-row-short matrix: detected and re-packed at next forward
    layer.set_weights("C2", torch.ones(3, 8))
    g = load_golden("model_deepfm.npz")
    batch = {k: v.clone() for k, v in g["batch"].items()}
    batch["C2"] = batch["C2"].clamp(max=2)
    m(batch)
    assert layer.arena.shape[0] == sum(v + 1 for v in (7, 3, 50, 11, 2)) - 1
    assert torch.equal(layer.embedding_layer["C2"].weight, torch.ones(3, 8))


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_lookup_pooled_on_cpu_matches_reference(case):
    """EmbeddingLayer.lookup_pooled + the drop-in MaskedSumPooling / MaskedAveragePooling modules on CPU (BASELINE config
    0) against the reference's `_seq` lookup + pooling (tests/golden/pool.npz), dense and CSR bags."""
    from conftest import load_golden
    from test_oracle_golden import pool_case
    from rec_pangu_amd.models.layers import EmbeddingLayer, MaskedAveragePooling, MaskedSumPooling
    table, seq, modes = pool_case(load_golden("pool.npz"), case)
    enc = {"C1": {"vocab_size": 7}, "I1": {"min": 0.0, "max": 1.0}, "hist": {"vocab_size": 60}, "C2": {"vocab_size": 3}}
    emb = EmbeddingLayer(enc, table.shape[1])
    with torch.no_grad():
        emb.embedding_layer["hist"].weight.copy_(table)
    X = {"hist_seq": seq}
    for mode, (out, cot, grad) in modes.items():
        pooling = "sum" if mode == "sum" else "average"
        emb.zero_grad()
        y = emb.lookup_pooled(X, "hist_seq", pooling)
        assert torch.equal(y, out)
        (y * cot).sum().backward()
        torch.testing.assert_close(emb.embedding_layer["hist"].weight.grad, grad, rtol=1e-6, atol=0)
        mod = MaskedSumPooling() if mode == "sum" else MaskedAveragePooling()
        assert torch.equal(mod(emb(X, name="hist_seq")), out)
        if case != "a":  # CSR bags without the (all-zero) padding ids
            keep = seq != 0
            offsets = torch.cat([torch.zeros(1, dtype=torch.long), keep.sum(1).cumsum(0)])
            y = emb.lookup_pooled({"hist_seq": seq[keep]}, "hist_seq", pooling, offsets=offsets)
            torch.testing.assert_close(y, out, rtol=1e-6, atol=1e-7)


def test_deepcopy_keeps_the_tables_in_one_arena():
    """copy.deepcopy(model) must not tear the table Parameters off the arena (Parameter.__deepcopy__ clones each one):
    the clone's tables are views of the clone's own arena, with the original's values, and independent of it."""
    import copy
    from rec_pangu_amd.models.ranking import DeepFM
    enc = {"I1": {"min": 0.0, "max": 1.0}, "C1": {"vocab_size": 7}, "C2": {"vocab_size": 30}}
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=8, hidden_units=[8], enc_dict=enc)
    twin = copy.deepcopy(model)
    lay, tl = model.embedding_layer, twin.embedding_layer
    assert tl.arena.data_ptr() != lay.arena.data_ptr() and torch.equal(tl.arena, lay.arena)
    off = 0
    for c in lay.emb_feature:
        w = tl.embedding_layer[c].weight
        assert w.data_ptr() == tl.arena.data_ptr() + off * 8 * 4, c
        assert w._rp_store() is tl
        off += w.shape[0]
    with torch.no_grad():
        tl.arena.add_(1.0)
    assert torch.equal(tl.embedding_layer["C2"].weight, lay.embedding_layer["C2"].weight + 1.0)
    b = {"I1": torch.rand(4), "C1": torch.tensor([0, 1, 7, 3]), "C2": torch.tensor([5, 30, 0, 2]), "label": torch.ones(4)}
    assert not torch.equal(twin(b)["pred"], model(b)["pred"])
