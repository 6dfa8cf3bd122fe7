"""GPU parity of the drop-in models (HIP path) against (a) the golden vectors captured from the
reference and (b) the CPU oracle at a mid-size Criteo-shaped batch.  Bar: logits/loss within 1e-4 of
the fp32 reference (BASELINE.json north_star); gradients and post-Adam weights within 1e-4 relative."""
import pytest
import torch

from conftest import load_golden, require_gpu, small_enc_dict
from oracle import ref_ops as R
from test_host_models import CASES, build

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


@pytest.mark.parametrize("name", list(CASES))
def test_model_forward_backward_vs_reference(name):
    from rec_pangu_amd import hip
    g = load_golden(f"model_{name}.npz")
    train_mode = CASES[name][1]
    model = build(name).to(DEV)
    model.train(train_mode)
    n0 = hip.launch_count()
    out = model(_to_dev(g["batch"]))
    assert hip.launch_count() > n0, "the HIP kernels did not run"
    for k, v in g["out"].items():
        torch.testing.assert_close(out[k].detach().cpu(), v, rtol=1e-4, atol=1e-5, msg=lambda m: f"{name}:{k}: {m}")
    model.zero_grad()
    out["loss"].backward()
    params = dict(model.named_parameters())
    for k, v in g["grad"].items():
        got = params[k].grad
        got = torch.zeros_like(v) if got is None else got.cpu()
        tol = 1e-4 * max(1e-2, float(v.abs().max()))
        assert (got - v).abs().max() <= tol, f"{name}: grad {k} off by {(got - v).abs().max()} (tol {tol})"


@pytest.mark.parametrize("name", ["deepfm", "fm", "wdl", "nfm", "dcn", "xdeepfm", "autoint_h2", "mmoe_train", "omoe_train",
                                  "mlmmoe_train", "sharebottom_train"])
def test_two_fused_adam_steps_vs_reference(name):
    from rec_pangu_amd.optim import make_adam, FusedAdam
    g = load_golden(f"model_{name}.npz")
    model = build(name).to(DEV)
    model.train(CASES[name][1])
    opt = make_adam(model, 1e-2)
    assert isinstance(opt, FusedAdam)
    for _ in range(2):
        r = model(_to_dev(g["batch"]))
        r["loss"].backward()
        opt.step()
        model.zero_grad()
    sd = model.state_dict()
    for k, v in g["adam2"].items():
        # Adam normalises the gradient, so a parameter whose true gradient is zero (a Linear bias in front
        # of a train-mode BatchNorm) moves by +-lr on pure rounding noise: not comparable across devices.
        # BatchNorm running statistics inherit that noise through the bias.
        if (k in g["grad"] and float(g["grad"][k].abs().max()) < 1e-6) or "running_" in k or "num_batches" in k:
            continue
        if v.dtype.is_floating_point:
            tol = 2e-4 * max(1e-2, float(v.abs().max()))
            assert (sd[k].cpu() - v).abs().max() <= tol, f"{name}: {k} off by {(sd[k].cpu() - v).abs().max()}"
    model.eval()
    with torch.no_grad():
        r = model(_to_dev(g["batch"]), is_training=False)
    # (mmoe_train: the +-lr noise steps of the pre-BatchNorm biases move the running means, hence eval outputs)
    atol = 2e-2 if name.endswith("_train") else 1e-4
    for k, v in g["adam2_out"].items():
        torch.testing.assert_close(r[k].cpu(), v, rtol=1e-3, atol=atol)


def test_grad_accumulation_and_zero_grad_semantics():
    """Two backward passes without zero_grad accumulate; zero_grad(set_to_none) then gives a fresh
    gradient with untouched rows exactly zero (the sparse re-zero invariant of the gradient arena)."""
    g = load_golden("model_deepfm.npz")
    model = build("deepfm").to(DEV)
    batch = _to_dev(g["batch"])
    model(batch)["loss"].backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters()}
    model(batch)["loss"].backward()
    for k, p in model.named_parameters():
        torch.testing.assert_close(p.grad, 2 * g1[k], rtol=1e-5, atol=1e-7)
    model.zero_grad()
    assert all(p.grad is None for p in model.parameters())
    b2 = {k: v.clone() for k, v in batch.items()}
    b2["C3"] = torch.zeros_like(b2["C3"])  # only row 0 of table C3 is looked up now
    model(b2)["loss"].backward()
    gC3 = dict(model.named_parameters())["embedding_layer.embedding_layer.C3.weight"].grad
    assert torch.count_nonzero(gC3[1:]) == 0 and torch.count_nonzero(gC3[0]) > 0


def test_index_out_of_range_raises_like_the_reference():
    g = load_golden("model_deepfm.npz")
    model = build("deepfm").to(DEV)
    batch = _to_dev(g["batch"])
    batch["C2"] = batch["C2"].clone()
    batch["C2"][5] = 4
    with pytest.raises(IndexError):
        model(batch)
    model(_to_dev(g["batch"]))  # flag was cleared


def test_deepfm_criteo_shape_midsize_vs_oracle():
    """26 sparse (Criteo cardinalities / 64) + 13 dense, D=64, MLP [64,64,64], B=4096: pred/loss and all
    gradients against the CPU oracle on the same weights."""
    card = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
            10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    enc = {f"I{i + 1}": {"min": 0.0, "max": 1.0} for i in range(13)}
    enc.update({f"C{i + 1}": {"vocab_size": max(2, c // 64)} for i, c in enumerate(card)})
    from rec_pangu_amd.models.ranking import DeepFM
    torch.manual_seed(0)
    model = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    sd = {k: v.clone().requires_grad_(True) for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(1)
    B = 4096
    batch = {f"I{i + 1}": torch.rand(B, generator=gen) for i in range(13)}
    batch.update({f"C{i + 1}": torch.randint(0, enc[f"C{i + 1}"]["vocab_size"] + 1, (B,), generator=gen)
                  for i in range(26)})
    batch["label"] = (torch.rand(B, generator=gen) < 0.25).float()
    ref = R.deepfm(sd, enc, batch)
    ref["loss"].backward()
    model = model.to(DEV)
    out = model(_to_dev(batch))
    out["loss"].backward()
    # at this shape the FM part of the embedding gradient rides in the first Linear's dgrad (rp_linear_fwd_rowadd)
    assert model.embedding_layer._fm_link is not None and model.embedding_layer._fm_link.folded
    torch.testing.assert_close(out["pred"].cpu(), ref["pred"].detach(), rtol=0, atol=1e-4)
    torch.testing.assert_close(out["loss"].cpu(), ref["loss"].detach(), rtol=0, atol=1e-4)
    for k, p in model.named_parameters():
        rg = sd[k].grad
        tol = 1e-4 * max(1e-4, float(rg.abs().max()))
        assert (p.grad.cpu() - rg).abs().max() <= tol, f"grad {k}: {(p.grad.cpu() - rg).abs().max()} > {tol}"


def test_sharded_layer_hip_primitives_single_rank():
    """The HIP side of the row-sharded path (local gather, rows->x+FM, routed gradient reduce) with a
    1-rank RCCL group: must equal the unsharded HIP model and the reference's golden output."""
    import torch.distributed as dist
    from rec_pangu_amd.sharded import shard_model_tables, allreduce_dense_grads, ShardedEmbeddingLayer
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        for name in ("deepfm", "xdeepfm", "fm"):
            g = load_golden(f"model_{name}.npz")
            model = build(name).to(DEV)
            model.train(CASES[name][1])
            model = shard_model_tables(model, 1, 0)
            out = model(_to_dev(g["batch"]))
            torch.testing.assert_close(out["pred"].detach().cpu(), g["out"]["pred"], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(out["loss"].detach().cpu(), g["out"]["loss"], rtol=1e-4, atol=1e-5)
            out["loss"].backward()
            allreduce_dense_grads(model)
            for lname, m in model.named_modules():
                if isinstance(m, ShardedEmbeddingLayer):
                    ref = torch.cat([g["grad"][f"{lname}.embedding_layer.{c}.weight"] for c in m.emb_feature])
                    got = m.local_arena.grad.cpu()
                    tol = 1e-4 * max(1e-2, float(ref.abs().max()))
                    assert (got - ref).abs().max() <= tol, f"{name}/{lname}: {(got - ref).abs().max()}"
            for k, p in model.named_parameters():
                if k in g["grad"]:
                    tol = 1e-4 * max(1e-2, float(g["grad"][k].abs().max()))
                    assert (p.grad.cpu() - g["grad"][k]).abs().max() <= tol, k
            # two optimiser steps (exact lazy dense Adam on the local shard) == the reference's Adam run
            from rec_pangu_amd.optim import make_adam
            model = shard_model_tables(build(name).to(DEV), 1, 0)
            model.train(CASES[name][1])
            opt = make_adam(model, 1e-2)
            for _ in range(2):
                model(_to_dev(g["batch"]))["loss"].backward()
                allreduce_dense_grads(model)
                opt.step()
                model.zero_grad()
            for lname, m in model.named_modules():
                if isinstance(m, ShardedEmbeddingLayer):
                    for col, tab in m.full_tables().items():
                        ref = g["adam2"][f"{lname}.embedding_layer.{col}.weight"]
                        tol = 2e-4 * max(1e-2, float(ref.abs().max()))
                        assert (tab.cpu() - ref).abs().max() <= tol, f"{name}/{lname}/{col} after 2 steps"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["deepfm", "xdeepfm", "dcn", "autoint"])
def test_full_size_split_batch_property(name):
    """BASELINE.json's full batch (65536 samples, 26 Criteo-shaped fields + 13 dense, D = 64; vocabularies / 16 to keep
    the test light) has no CPU oracle run — instead a size-independent property of these models (no BatchNorm): the
    samples are independent, so the full-batch predictions equal those of its two halves run separately, the loss is
    their mean, and the gradients are the mean of the halves' gradients (linearity of the backward)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    enc = bench.criteo_enc_dict(16)
    torch.manual_seed(7)
    model = bench.build_model(name, enc).to(DEV)
    model.eval()  # Dropout off (these models have no BatchNorm); the backward below is the training backward
    B = 65536
    full = bench.synth_batch(enc, B, 11, DEV)
    halves = [{k: v[:B // 2] for k, v in full.items()}, {k: v[B // 2:] for k, v in full.items()}]

    def run(batch):
        model.zero_grad(set_to_none=True)
        out = model(batch)
        out["loss"].backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        return out["pred"].detach().clone(), float(out["loss"].detach()), grads

    pf, lf, gf = run(full)
    p0, l0, g0 = run(halves[0])
    p1, l1, g1 = run(halves[1])
    torch.testing.assert_close(pf, torch.cat([p0, p1]), rtol=1e-5, atol=1e-6)
    assert abs(lf - 0.5 * (l0 + l1)) <= 1e-5 * max(1.0, abs(lf))
    assert set(gf) == set(g0) == set(g1)
    for k in gf:
        ref = 0.5 * (g0[k] + g1[k])
        tol = 2e-4 * max(1e-6, float(ref.abs().max()))
        assert float((gf[k] - ref).abs().max()) <= tol, f"{name}: {k}: {float((gf[k] - ref).abs().max())} > {tol}"


def test_sync_batchnorm_on_hip_single_rank():
    """SyncBatchNorm1d (what the towers use on G > 1 ranks) on HIP tensors under a 1-rank RCCL group: with one rank the
    global batch is the local batch, so the MMOE step must equal the plain HIP model's (rp_batchnorm_* kernels)."""
    import socket
    import torch.distributed as dist
    from rec_pangu_amd.sharded import sync_batchnorm, SyncBatchNorm1d
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        g = load_golden("model_mmoe_train.npz")
        outs = []
        for sync in (False, True):
            model = build("mmoe_train").to(DEV)
            model.train(CASES["mmoe_train"][1])
            if sync:
                sync_batchnorm(model)
                assert any(isinstance(m, SyncBatchNorm1d) for m in model.modules())
            out = model(_to_dev(g["batch"]))
            out["loss"].backward()
            outs.append((out, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                         {k: v.detach().clone() for k, v in model.named_buffers() if "running" in k}))
        for k in ("task1_pred", "task2_pred", "loss"):
            torch.testing.assert_close(outs[1][0][k].detach(), outs[0][0][k].detach(), rtol=1e-5, atol=1e-6)
        for k, v in outs[0][1].items():
            tol = 1e-4 * max(1e-2, float(v.abs().max()))  # (pre-BatchNorm biases: both sides are ~1e-8 rounding noise)
            assert float((outs[1][1][k] - v).abs().max()) <= tol, k
        for k, v in outs[0][2].items():
            torch.testing.assert_close(outs[1][2][k], v, rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()
