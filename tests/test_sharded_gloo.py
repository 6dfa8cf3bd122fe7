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


def _worker(rank, world, port, name, how, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd.sharded import (shard_model_tables, build_sharded_model, allreduce_dense_grads,
                                           ShardedEmbeddingLayer)
        g = load_golden(f"model_{name}.npz")
        B = g["batch"]["label"].shape[0]
        b = B // world
        local = {k: v[rank * b:(rank + 1) * b].clone() for k, v in g["batch"].items()}
        if how == "local":  # shard-local construction: the full arena never exists on any rank
            model = build_sharded_model(lambda: _build(name), world, rank)
            assert not any(type(m).__name__ == "EmbeddingLayer" for m in model.modules())
        else:               # cut from a fully built model
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
        # out-of-range ids are reported like the single-process path — on EVERY rank, also when only one rank's batch
        # holds the bad id (the flag is OR-ed over the ranks: a rank that kept going alone would hang its peers)
        bad = {k: v.clone() for k, v in local.items()}
        if rank == 0:
            bad["C2"][0] = 99
        try:
            model(bad)
            res["raised"] = False
        except IndexError:
            res["raised"] = True
        ret[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,how", [("deepfm", "cut"), ("xdeepfm", "cut"), ("deepfm", "local"), ("xdeepfm", "local")])
def test_two_ranks_equal_single_process(name, how):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), name, how, ret), nprocs=world, join=True)
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


def _build_mmoe():
    from rec_pangu_amd.models.multi_task import MMOE
    from conftest import small_enc_dict
    torch.manual_seed(4321)
    m = MMOE(num_task=2, n_expert=3, embedding_dim=8, mmoe_hidden_dim=12, hidden_dim=[10, 6], dropouts=[0.0, 0.0],
             enc_dict=small_enc_dict())
    m.train()  # BatchNorm in training mode: batch statistics
    return m


def _mmoe_batch(B=32):
    g = torch.Generator().manual_seed(99)
    from conftest import small_enc_dict
    batch = {}
    for k, v in small_enc_dict().items():
        if "vocab_size" in v:
            batch[k] = torch.randint(0, v["vocab_size"] + 1, (B,), generator=g)
        else:
            batch[k] = torch.rand(B, generator=g)
    batch["task1_label"] = (torch.rand(B, generator=g) < 0.4).float()
    batch["task2_label"] = (torch.rand(B, generator=g) < 0.2).float()
    return batch


def _mmoe_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd.sharded import shard_model_tables, allreduce_dense_grads, SyncBatchNorm1d
        batch = _mmoe_batch()
        b = batch["task1_label"].shape[0] // world
        local = {k: v[rank * b:(rank + 1) * b].clone() for k, v in batch.items()}
        model = shard_model_tables(_build_mmoe(), world, rank)
        assert any(isinstance(m, SyncBatchNorm1d) for m in model.modules())
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        res = {}
        for step in range(2):
            out = model(local)
            out["loss"].backward()
            allreduce_dense_grads(model)
            if step == 0:
                res["pred"] = [out[f"task{i + 1}_pred"].detach().clone() for i in range(2)]
                res["loss"] = out["loss"].detach().clone()
                res["grads"] = {k: p.grad.clone() for k, p in model.named_parameters() if "local_arena" not in k}
                res["buffers"] = {k: v.clone() for k, v in model.named_buffers() if "running" in k}
            opt.step()
            model.zero_grad()
        res["dense2"] = {k: p.detach().clone() for k, p in model.named_parameters() if "local_arena" not in k}
        ret[rank] = res
    finally:
        dist.destroy_process_group()


def test_mmoe_batchnorm_two_ranks_equal_single_process():
    """MMOE's towers carry BatchNorm1d in training mode: with SyncBatchNorm1d (global-batch statistics, swapped in by
    shard_model_tables) two ranks reproduce the single-process run on the global batch — predictions, loss, dense
    gradients, weights after two Adam steps and the running statistics."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mmoe_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    batch = _mmoe_batch()
    model = _build_mmoe()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for step in range(2):
        out = model({k: v.clone() for k, v in batch.items()})
        out["loss"].backward()
        if step == 0:
            for i in range(2):
                pred = torch.cat([ret[r]["pred"][i] for r in range(world)])
                torch.testing.assert_close(pred, out[f"task{i + 1}_pred"].detach(), rtol=1e-5, atol=1e-6)
            mean_loss = sum(ret[r]["loss"] for r in range(world)) / world
            torch.testing.assert_close(mean_loss, out["loss"].detach(), rtol=1e-5, atol=1e-6)
            ref = {k: p.grad.clone() for k, p in model.named_parameters()}
            for k, gk in ret[0]["grads"].items():
                torch.testing.assert_close(gk, ref[k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
            # running statistics after the first step (later ones see the noise-driven pre-BatchNorm biases, below)
            bufs = dict(model.named_buffers())
            for k, v in ret[0]["buffers"].items():
                torch.testing.assert_close(v, bufs[k], rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
        opt.step()
        model.zero_grad()
    sd = dict(model.named_parameters())
    for k, v in ret[1]["dense2"].items():
        if float(ref[k].abs().max()) < 1e-6:
            # a bias in front of a BatchNorm (ctr_hidden_j.bias, and ctr_batchnorm_j.bias when another Linear + BatchNorm
            # follows) has an exactly-zero true gradient; what is left is rounding noise, which Adam normalises to
            # +-lr steps whose sign depends on the summation order: not comparable (2 steps * lr bound only)
            assert (v - sd[k].detach()).abs().max() <= 2 * 2 * 1e-2 + 1e-6
            continue
        torch.testing.assert_close(v, sd[k].detach(), rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")


# ---- sharded checkpoints in the reference layout, with optimizer state ---------------------------------
def _ckpt_worker(rank, world, port, ckpt_dir, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd.sharded import build_sharded_model, allreduce_dense_grads, ShardedEmbeddingLayer
        from rec_pangu_amd.checkpoint import save_checkpoint, load_checkpoint
        from rec_pangu_amd.trainer import RankTrainer
        name = "xdeepfm"  # two sharded layers (D-wide tables + the LR_Layer's 1-wide ones)
        g = load_golden(f"model_{name}.npz")
        B = g["batch"]["label"].shape[0]
        b = B // world
        local = {k: v[rank * b:(rank + 1) * b].clone() for k, v in g["batch"].items()}
        other = {k: (v.flip(0) if v.dtype.is_floating_point else (v * 0 + 1)) for k, v in local.items()}

        def make(seed_shift=0):
            m = build_sharded_model(lambda: _build(name), world, rank)
            if seed_shift:  # different starting weights: a load must overwrite everything
                with torch.no_grad():
                    for p in m.parameters():
                        p.add_(0.1 * seed_shift)
            # (_build leaves xDeepFM in eval mode: its MLP's default dropout would make two runs incomparable)
            return m, torch.optim.Adam(m.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)

        def train(m, opt, batches):
            for bt in batches:
                m(bt)["loss"].backward()
                allreduce_dense_grads(m)
                opt.step()
                m.zero_grad()

        def snapshot(m):
            with torch.no_grad():
                pred = m(local, is_training=False)["pred"].clone()
            return {"pred": pred, "params": {k: p.detach().clone() for k, p in m.named_parameters()}}

        seq = [local, other, other, local]
        m_ref, o_ref = make()
        train(m_ref, o_ref, seq)                      # the uninterrupted run
        m1, o1 = make()
        train(m1, o1, seq[:2])
        save_checkpoint(m1, small_enc_dict(), ckpt_dir, optimizer=o1)
        mid = snapshot(m1)
        m2, o2 = make(seed_shift=1)                   # "new process": different weights, fresh optimizer
        extra = load_checkpoint(m2, ckpt_dir, optimizer=o2)
        assert extra["enc_dict"] == small_enc_dict()
        loaded = snapshot(m2)
        for k in mid["params"]:
            assert torch.equal(mid["params"][k], loaded["params"][k]), f"load: {k}"
        train(m2, o2, seq[2:])
        a, c = snapshot(m_ref), snapshot(m2)
        for k in a["params"]:
            assert torch.equal(a["params"][k], c["params"][k]), f"resume differs from the uninterrupted run: {k}"
        assert torch.equal(a["pred"], c["pred"])
        # RankTrainer.save_all on a sharded model writes the reference layout too (rank 0 merges)
        RankTrainer(num_task=1).save_all(m_ref, small_enc_dict(), os.path.join(ckpt_dir, "final"))
        ret[rank] = {"mid_pred": mid["pred"], "final_pred": a["pred"],
                     "n_sharded": sum(isinstance(x, ShardedEmbeddingLayer) for x in m_ref.modules())}
    finally:
        dist.destroy_process_group()


def test_sharded_checkpoint_reference_layout_resume_and_reshard(tmp_path):
    """world-2 save -> (a) the merged model.pth has exactly the reference's state_dict keys and loads into a
    single-process model whose predictions on the global batch equal the two ranks' predictions; (b) save / reload /
    continue equals the uninterrupted run bit for bit (weights, Adam moments, step count: asserted in the workers);
    (c) the same files re-shard to another world size (1) and the optimizer file carries per-table moments."""
    from rec_pangu_amd.checkpoint import load_checkpoint
    from rec_pangu_amd.sharded import build_sharded_model
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    ckpt_dir = str(tmp_path / "ck")
    mp.spawn(_ckpt_worker, args=(world, _free_port(), ckpt_dir, ret), nprocs=world, join=True)
    assert len(ret) == world and ret[0]["n_sharded"] == 2
    g = load_golden("model_xdeepfm.npz")
    # (the per-rank shard files are removed after the merge: ADVICE r2)
    assert sorted(f for f in os.listdir(ckpt_dir) if f.endswith(".pth")) == ["model.pth", "optimizer.pth"]
    saved = torch.load(os.path.join(ckpt_dir, "model.pth"), weights_only=False)
    assert sorted(saved.keys()) == ["enc_dict", "model"]
    plain = _build("xdeepfm")
    assert list(saved["model"].keys()) == list(plain.state_dict().keys()), "reference state_dict keys, in order"
    plain.load_state_dict(saved["model"])  # examples/ranking/inference_example.py:29-37
    plain.eval()
    with torch.no_grad():
        pred = plain({k: v.clone() for k, v in g["batch"].items()}, is_training=False)["pred"]
    torch.testing.assert_close(torch.cat([ret[r]["mid_pred"] for r in range(world)]), pred, rtol=1e-5, atol=1e-6)
    final = torch.load(os.path.join(ckpt_dir, "final", "model.pth"), weights_only=False)
    plain.load_state_dict(final["model"])
    with torch.no_grad():
        pred = plain({k: v.clone() for k, v in g["batch"].items()}, is_training=False)["pred"]
    torch.testing.assert_close(torch.cat([ret[r]["final_pred"] for r in range(world)]), pred, rtol=1e-5, atol=1e-6)
    # (c) re-shard to world 1 (no process group needed to load), moments included
    one = build_sharded_model(lambda: _build("xdeepfm"), 1, 0)
    opt = torch.optim.Adam(one.parameters(), lr=1e-2)
    load_checkpoint(one, ckpt_dir, optimizer=opt)
    osd = torch.load(os.path.join(ckpt_dir, "optimizer.pth"), weights_only=False)
    assert osd["step"] == 2
    for lname in ("embedding_layer", "lr_layer.emb_layer"):
        lay = one.get_submodule(lname)
        full = torch.cat([saved["model"][f"{lname}.embedding_layer.{c}.weight"] for c in lay.emb_feature])
        assert torch.equal(lay.local_arena.detach(), full)
        m_full = torch.cat([osd["state"][f"{lname}.embedding_layer.{c}.weight"]["exp_avg"] for c in lay.emb_feature])
        assert torch.equal(opt.state[lay.local_arena]["exp_avg"], m_full) and float(m_full.abs().max()) > 0


def _wire_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd import sharded
        from rec_pangu_amd.sharded import build_sharded_model, allreduce_dense_grads
        g = load_golden("model_deepfm.npz")
        b = g["batch"]["label"].shape[0] // world
        local = {k: v[rank * b:(rank + 1) * b].clone() for k, v in g["batch"].items()}
        res = {}
        for wire in (torch.float32, torch.bfloat16):
            model = build_sharded_model(lambda: _build("deepfm"), world, rank)
            for m in model.modules():
                if hasattr(m, "wire_dtype"):
                    m.wire_dtype = wire
            sharded._WIRE_BYTES[0] = 0
            out = model(local)
            out["loss"].backward()
            allreduce_dense_grads(model)
            res[str(wire)] = {"pred": out["pred"].detach().clone(), "bytes": sharded._WIRE_BYTES[0],
                              "grad": model.embedding_layer.local_arena.grad.clone(),
                              "dense": {k: p.grad.clone() for k, p in model.named_parameters() if "local_arena" not in k}}
        ret[rank] = res
    finally:
        dist.destroy_process_group()


def test_bf16_wire_mode_two_ranks():
    """ShardedEmbeddingLayer.wire_dtype = torch.bfloat16: the looked-up rows and their gradients travel as bf16 — HALF the
    bytes of both row exchanges — and come out within the stated tolerance of the fp32 wire (2^-9 relative per travelling
    value: predictions within 1e-2, gradients within 1 % of their scale); the fp32 wire stays the default / parity mode."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_wire_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        f32, b16 = ret[r]["torch.float32"], ret[r]["torch.bfloat16"]
        assert f32["bytes"] > 0 and b16["bytes"] * 2 == f32["bytes"]
        dp = float((f32["pred"] - b16["pred"]).abs().max())
        assert 0.0 < dp <= 1e-2, dp
        assert float((f32["grad"] - b16["grad"]).abs().max()) <= 1e-2 * float(f32["grad"].abs().max())
        for k, v in f32["dense"].items():
            assert float((v - b16["dense"][k]).abs().max()) <= 2e-2 * max(1e-6, float(v.abs().max())), k


# ---- a REPLICATED model saved from a multi-rank job ----------------------------------------------------------
def _replicated_ckpt_worker(rank, world, port, ckpt_dir, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    torch.set_num_threads(1)
    try:
        from rec_pangu_amd.checkpoint import save_checkpoint, load_checkpoint
        torch.manual_seed(0)
        m = _build("deepfm")  # the same weights on every rank (what DDP-style replicas hold)
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
        g = load_golden("model_deepfm.npz")
        m.train()
        m(g["batch"])["loss"].backward()
        opt.step()
        m.zero_grad()
        # (a) the single-process habit: rank 0 alone saves.  Must neither hang nor need the other ranks.
        if rank == 0:
            save_checkpoint(m, small_enc_dict(), os.path.join(ckpt_dir, "a"), optimizer=opt)
        dist.barrier()
        # (b) every rank calls it without asking for a barrier: one writer, nobody waits
        save_checkpoint(m, small_enc_dict(), os.path.join(ckpt_dir, "b"), optimizer=opt)
        dist.barrier()
        # (c) collective=True: every rank returns only once the file is complete
        save_checkpoint(m, small_enc_dict(), os.path.join(ckpt_dir, "c"), optimizer=opt, collective=True)
        assert os.path.exists(os.path.join(ckpt_dir, "c", "model.pth"))
        out = {}
        for d in ("a", "b", "c"):
            torch.manual_seed(1)
            m2 = _build("deepfm")
            load_checkpoint(m2, os.path.join(ckpt_dir, d))
            out[d] = all(torch.equal(p, q) for p, q in zip(m.state_dict().values(), m2.state_dict().values()))
            out[d + "_files"] = sorted(f for f in os.listdir(os.path.join(ckpt_dir, d)) if f.endswith(".pth"))
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_replicated_model_checkpoint_from_two_ranks(tmp_path):
    """save_checkpoint of an UNSHARDED model inside a 2-rank job: `if rank == 0: save_checkpoint(...)` works (no hidden
    barrier: round 4's version hung rank 0 until the process-group timeout), "every rank calls it" has one writer and no
    leftover shard files, collective=True adds the barrier; all three load back to the same weights."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    ckpt_dir = str(tmp_path / "ck")
    mp.spawn(_replicated_ckpt_worker, args=(world, _free_port(), ckpt_dir, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for d in ("a", "b", "c"):
            assert ret[r][d], f"rank {r}: checkpoint {d} did not restore the weights"
            assert ret[r][d + "_files"] == ["model.pth", "optimizer.pth"]
