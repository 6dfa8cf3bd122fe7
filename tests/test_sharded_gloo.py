"""N>1 path on CPU: world_size 2, gloo.  Row-sharded tables + all-to-all lookup must reproduce the
single-process model on the global batch: predictions, loss, dense gradients (after the flat all-reduce),
table gradients, and the weights after two Adam steps (SURVEY.md §8e equivalence test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, small_enc_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(name):
    from rec_pangu_amd.models.ranking import DeepFM, xDeepFM
    enc = small_enc_dict()
    torch.manual_seed(1234)
    if name == "deepfm":
        return DeepFM(embedding_dim=8, hidden_units=[16, 8], enc_dict=enc)
    m = xDeepFM(embedding_dim=8, dnn_hidden_units=[16, 8], cin_layer_units=[6, 4], enc_dict=enc)
    m.eval()
    return m


def _worker(rank, world, port, name, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd.sharded import shard_model_tables, allreduce_dense_grads, ShardedEmbeddingLayer
        g = load_golden(f"model_{name}.npz")
        B = g["batch"]["label"].shape[0]
        b = B // world
        local = {k: v[rank * b:(rank + 1) * b].clone() for k, v in g["batch"].items()}
        model = shard_model_tables(_build(name), world, rank)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
        out = model(local)
        out["loss"].backward()
        allreduce_dense_grads(model)
        res = {"pred": out["pred"].detach().clone(), "loss": out["loss"].detach().clone()}
        res["dense_grads"] = {k: p.grad.clone() for k, p in model.named_parameters() if "local_arena" not in k}
        layers = {n: m for n, m in model.named_modules() if isinstance(m, ShardedEmbeddingLayer)}
        res["local_grads"] = {n: m.local_arena.grad.clone() for n, m in layers.items()}
        opt.step()
        model.zero_grad()
        out = model(local)
        out["loss"].backward()
        allreduce_dense_grads(model)
        opt.step()
        model.zero_grad()
        res["tables2"] = {n: m.full_tables() for n, m in layers.items()}
        res["dense2"] = {k: p.detach().clone() for k, p in model.named_parameters() if "local_arena" not in k}
        # out-of-range ids are reported like the single-process path
        bad = {k: v.clone() for k, v in local.items()}
        bad["C2"][0] = 99
        try:
            model(bad)
            res["raised"] = False
        except IndexError:
            res["raised"] = True
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["deepfm", "xdeepfm"])
def test_two_ranks_equal_single_process(name):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), name, ret), nprocs=world, join=True)
    assert len(ret) == world
    g = load_golden(f"model_{name}.npz")
    # single-process run on the global batch
    model = _build(name)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    out = model({k: v.clone() for k, v in g["batch"].items()})
    out["loss"].backward()
    pred = torch.cat([ret[r]["pred"] for r in range(world)])
    torch.testing.assert_close(pred, out["pred"].detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pred, g["out"]["pred"], rtol=1e-5, atol=1e-6)  # == the reference's own output
    mean_loss = sum(ret[r]["loss"] for r in range(world)) / world
    torch.testing.assert_close(mean_loss, out["loss"].detach(), rtol=1e-5, atol=1e-6)
    ref_grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    for k, gk in ret[0]["dense_grads"].items():
        torch.testing.assert_close(gk, ref_grads[k], rtol=1e-4, atol=1e-7, msg=lambda m: f"{k}: {m}")
        assert torch.equal(gk, ret[1]["dense_grads"][k]), "replicated gradients must be identical after all-reduce"
    for lname in ret[0]["local_grads"]:
        layer = model.get_submodule(lname)
        full = torch.cat([layer.embedding_layer[c].weight.grad for c in layer.emb_feature])
        for r in range(world):
            torch.testing.assert_close(ret[r]["local_grads"][lname], full[r::world], rtol=1e-4, atol=1e-7)
    opt.step()
    model.zero_grad()
    model({k: v.clone() for k, v in g["batch"].items()})["loss"].backward()
    opt.step()
    for lname, tables in ret[0]["tables2"].items():
        layer = model.get_submodule(lname)
        for c in layer.emb_feature:
            torch.testing.assert_close(tables[c], layer.embedding_layer[c].weight.detach(), rtol=1e-4, atol=1e-6)
    sd = dict(model.named_parameters())
    for k, v in ret[1]["dense2"].items():
        torch.testing.assert_close(v, sd[k].detach(), rtol=1e-4, atol=1e-6)
    assert all(ret[r]["raised"] for r in range(world))
