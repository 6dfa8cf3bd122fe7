"""GPU parity of the interaction-block kernels (CrossNet, MMOE combine, BatchNorm, field attention, CIN)
against the golden layer fixtures captured from the reference (tests/golden/layers.npz) and against the CPU
oracle on larger seeded inputs."""
import pytest
import torch

from conftest import load_golden, require_gpu
from oracle import ref_ops as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


def _close(a, b, rel=1e-4, floor=1e-3, what=""):
    tol = rel * max(floor, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol, f"{what}: max err {err} > {tol}"


# ------------------------------------------------------------------------------------------------ CrossNet
def test_crossnet_vs_reference_fixture():
    from rec_pangu_amd import functional as Fh
    c = load_golden("layers.npz")["cross"]
    W = torch.stack([c[f"w/cross_net.{i}.weight.weight"].reshape(-1) for i in range(3)]).to(DEV).requires_grad_(True)
    Bv = torch.stack([c[f"w/cross_net.{i}.bias"] for i in range(3)]).to(DEV).requires_grad_(True)
    x0 = c["in"].to(DEV).requires_grad_(True)
    y = Fh.crossnet(x0, W, Bv)
    _close(y.detach().cpu(), c["out"], what="cross out")
    (y * torch.linspace(0.5, 1.5, 43, device=DEV)).sum().backward()
    _close(x0.grad.cpu(), c["grad_in"], what="cross dx0")
    for i in range(3):
        _close(W.grad[i].cpu(), c[f"gw/cross_net.{i}.weight.weight"].reshape(-1), what=f"cross dW{i}")
        _close(Bv.grad[i].cpu(), c[f"gw/cross_net.{i}.bias"], what=f"cross dB{i}")


@pytest.mark.parametrize("B,d,ld,L", [(1000, 1677, 1696, 3), (257, 649, 672, 1), (64, 2048, 2048, 6), (33, 5, 5, 2)])
def test_crossnet_fused_fc_vs_oracle(B, d, ld, L):
    from rec_pangu_amd import functional as Fh
    g = torch.Generator().manual_seed(B + d)
    x = torch.zeros(B, ld)
    x[:, :d] = torch.randn(B, d, generator=g)
    W = (torch.randn(L, d, generator=g) / d ** 0.5)
    Bv = torch.randn(L, d, generator=g) * 0.1
    wfc = torch.randn(1, d, generator=g) / d ** 0.5
    bfc = torch.randn(1, generator=g)
    coef = torch.randn(B, 1, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x[:, :d], W, Bv, wfc, bfc)]
    xl = R.cross_net(ref_in[0], [w.reshape(1, -1) for w in ref_in[1]], list(ref_in[2]))
    ref_logit = xl @ ref_in[3].t() + ref_in[4]
    (ref_logit * coef).sum().backward()
    dev_in = [t.to(DEV).requires_grad_(True) for t in (x, W, Bv, wfc, bfc)]
    logit = Fh.crossnet(*dev_in)
    _close(logit.detach().cpu(), ref_logit.detach(), what="logit")
    (logit * coef.to(DEV)).sum().backward()
    _close(dev_in[0].grad[:, :d].cpu(), ref_in[0].grad, what="dx0")
    assert torch.count_nonzero(dev_in[0].grad[:, d:]) == 0
    for i, name in ((1, "dW"), (2, "dB"), (3, "dwfc"), (4, "dbfc")):
        _close(dev_in[i].grad.cpu(), ref_in[i].grad, rel=2e-4, what=name)
    # unfused variant returns X_L itself
    xl_dev = Fh.crossnet(dev_in[0].detach(), dev_in[1].detach(), dev_in[2].detach())
    _close(xl_dev.cpu(), xl.detach(), what="X_L")


# ------------------------------------------------------------------------------------------------ CIN
def test_cin_vs_reference_fixture():
    """CompressedInteractionNet (two layers, the last one collapsed into fc) against the reference's own output and
    gradients (tests/golden/layers.npz cin/*)."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import CompressedInteractionNet
    c = load_golden("layers.npz")["cin"]
    cin = CompressedInteractionNet(5, [6, 4], output_dim=1)
    cin.load_state_dict({k[2:]: v for k, v in c.items() if k.startswith("w/")})
    cin = cin.to(DEV)
    x = c["in"].to(DEV).requires_grad_(True)
    n0 = hip.launch_count()
    y = cin(x)
    assert hip.launch_count() > n0
    _close(y.detach().cpu(), c["out"], what="cin out")
    (y.squeeze(-1) * torch.linspace(-1, 1, x.shape[0], device=DEV)).sum().backward()
    _close(x.grad.cpu(), c["grad_in"], rel=2e-4, what="cin dX0")
    for k, p in cin.named_parameters():
        # the fixture's output weights sum to zero, so the bias gradients are pure cancellation: absolute floor
        _close(p.grad.cpu(), c["gw/" + k], rel=2e-4, floor=1e-2, what=f"cin d{k}")


# [40, 70, 16] / [128, 128, 128]: middle layers fed by MORE than 32 maps — the chunked bf16 matrix-core form
# (functional._CINChunked: chunks of 32 + 8, 32 + 32 + 6, 4 x 32 maps), through the misaligned-row fallback here
@pytest.mark.parametrize("B,H,D,units", [(300, 26, 64, [128, 128]), (257, 26, 64, [16, 16, 16]), (130, 16, 40, [8, 8]),
                                          (64, 5, 8, [7]), (100, 26, 32, [32, 24, 9]), (300, 26, 64, [40, 70, 16]),
                                          (130, 26, 64, [128, 128, 128]), (96, 7, 32, [33, 5])])
def test_cin_vs_oracle(B, H, D, units):
    from rec_pangu_amd.models.layers import CompressedInteractionNet
    g = torch.Generator().manual_seed(B + H)
    torch.manual_seed(B)
    cin = CompressedInteractionNet(H, units, output_dim=1)
    ld = H * D + 13
    xbuf = torch.randn(B, ld, generator=g) * 0.5
    coef = torch.randn(B, 1, generator=g)
    rx = xbuf[:, :H * D].reshape(B, H, D).clone().requires_grad_(True)
    rw = {k: v.detach().clone().requires_grad_(True) for k, v in cin.named_parameters()}
    n = len(units)
    ref = R.cin(rx, [rw[f"cin_layer.layer_{i + 1}.weight"] for i in range(n)],
                [rw[f"cin_layer.layer_{i + 1}.bias"] for i in range(n)], rw["fc.weight"], rw["fc.bias"])
    (ref * coef).sum().backward()
    cin = cin.to(DEV)
    dxbuf = xbuf.to(DEV).requires_grad_(True)
    out = cin(dxbuf[:, :H * D].unflatten(1, (H, D)))  # strided view, as the models pass it
    _close(out.detach().cpu(), ref.detach(), rel=2e-4, what="cin out")
    (out * coef.to(DEV)).sum().backward()
    _close(dxbuf.grad[:, :H * D].cpu().reshape(B, H, D), rx.grad, rel=3e-4, what="cin dX0")
    for k, p in cin.named_parameters():
        _close(p.grad.cpu(), rw[k].grad, rel=3e-4, what=f"cin d{k}")


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("tag,din,heads,adim", [("a", 8, 2, 4), ("b", 8, 2, 3), ("c", 8, 1, 8), ("d", 6, 3, 5)])
def test_field_attention_vs_reference_fixture(tag, din, heads, adim):
    """MultiHeadSelfAttention (raw-view head split, no scale, W_res only when Din != H*a, ReLU) forward and
    all gradients against what the reference produced (tests/golden/layers.npz attn_*)."""
    from rec_pangu_amd.models.layers import MultiHeadSelfAttention
    c = load_golden("layers.npz")[f"attn_{tag}"]
    att = MultiHeadSelfAttention(din, attention_dim=adim, num_heads=heads, align_to="output")
    att.load_state_dict({k[2:]: v for k, v in c.items() if k.startswith("w/")})
    att = att.to(DEV)
    x = c["in"].to(DEV).requires_grad_(True)
    y = att(x)
    _close(y.detach().cpu(), c["out"], what="attention out")
    (y * y).sum().backward()
    _close(x.grad.cpu(), c["grad_in"], rel=2e-4, what="attention dX")
    for k, p in att.named_parameters():
        _close(p.grad.cpu(), c["gw/" + k], rel=2e-4, what=f"attention d{k}")


def test_field_attention_oversize_config_is_composed_not_crashed():
    """H*a = 64 at Din = 64 needs > 160 KB of LDS in the backward: the layer reports it and the module composes
    the same arithmetic from device ops instead (still on the GPU, still correct)."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import MultiHeadSelfAttention
    assert hip.field_attention_fits(26, 64, 1, 8, True) and hip.field_attention_fits(26, 64, 2, 16, True)
    assert not hip.field_attention_fits(26, 64, 2, 32, False)
    g = torch.Generator().manual_seed(0)
    att = MultiHeadSelfAttention(64, attention_dim=32, num_heads=2, align_to="output")
    x = torch.randn(50, 26, 64, generator=g)
    ref = R.mhsa(x, att.W_q.weight, att.W_k.weight, att.W_v.weight, None, 2, 32)
    n0 = hip.launch_count()
    out = att.to(DEV)(x.to(DEV))
    assert hip.launch_count() == n0
    _close(out.detach().cpu(), ref.detach(), what="composed attention")


@pytest.mark.parametrize("B,T,din,heads,adim,scale", [(1000, 26, 64, 1, 8, False), (513, 26, 64, 2, 16, True),
                                                       (300, 16, 40, 4, 10, False), (70, 39, 16, 4, 8, False)])
def test_field_attention_vs_oracle(B, T, din, heads, adim, scale):
    from rec_pangu_amd import functional as Fh
    g = torch.Generator().manual_seed(B + T)
    ld = T * din + 13
    x = torch.randn(B, ld, generator=g)
    HA = heads * adim
    has_res = din != HA
    ws = [torch.randn(HA, din, generator=g) / din ** 0.5 for _ in range(4 if has_res else 3)]
    coef = torch.randn(B, T, HA, generator=g)
    rx = x[:, :T * din].reshape(B, T, din).clone().requires_grad_(True)
    rw = [w.clone().requires_grad_(True) for w in ws]
    ref = R.mhsa(rx, rw[0], rw[1], rw[2], rw[3] if has_res else None, heads, adim, use_scale=scale)
    (ref * coef).sum().backward()
    dx = x.to(DEV).requires_grad_(True)
    dw = torch.cat(ws).to(DEV).requires_grad_(True)
    out = Fh.field_attention(dx, dw, T, din, heads, adim, has_res, adim ** 0.5 if scale else 0.0)
    _close(out.detach().cpu(), ref.detach(), what="out")
    (out * coef.to(DEV)).sum().backward()
    _close(dx.grad[:, :T * din].cpu().reshape(B, T, din), rx.grad, rel=2e-4, what="dX")
    assert torch.count_nonzero(dx.grad[:, T * din:]) == 0
    _close(dw.grad.cpu(), torch.cat([w.grad for w in rw]), rel=3e-4, what="dW")


@pytest.mark.parametrize("B,T,din,heads,adim,scale", [(1000, 26, 64, 1, 8, False), (513, 26, 64, 2, 16, True),
                                                       (300, 16, 40, 4, 10, False), (70, 39, 16, 4, 8, False),
                                                       (257, 26, 8, 1, 8, False), (65, 70, 12, 3, 5, True)])
def test_field_attention_split_form_vs_oracle(B, T, din, heads, adim, scale):
    """projection GEMM (rp_linear_fwd) + T x T core (rp_attention_core_*): output and every gradient vs the oracle,
    including H*T > 64 rows per sample, a non-power-of-two head width and the no-W_res case (Din == H*a)."""
    from rec_pangu_amd import functional as Fh, hip
    g = torch.Generator().manual_seed(B + T + din)
    HA = heads * adim
    has_res = din != HA
    assert hip.attention_core_fits(T, heads, adim)
    x = torch.randn(B, T, din, generator=g)
    ws = [torch.randn(HA, din, generator=g) / din ** 0.5 for _ in range(4 if has_res else 3)]
    coef = torch.randn(B, T, HA, generator=g)
    rx = x.clone().requires_grad_(True)
    rw = [w.clone().requires_grad_(True) for w in ws]
    ref = R.mhsa(rx, rw[0], rw[1], rw[2], rw[3] if has_res else None, heads, adim, use_scale=scale)
    (ref * coef).sum().backward()
    dx = x.to(DEV).requires_grad_(True)
    dw = torch.cat(ws).to(DEV).requires_grad_(True)
    n0 = hip.launch_count()
    out = Fh.field_attention_split(dx, dw, T, din, heads, adim, has_res, adim ** 0.5 if scale else 0.0)
    assert hip.launch_count() >= n0 + 2
    _close(out.detach().cpu(), ref.detach(), what="out")
    (out * coef.to(DEV)).sum().backward()
    _close(dx.grad.cpu(), rx.grad, rel=2e-4, what="dX")
    _close(dw.grad.cpu(), torch.cat([w.grad for w in rw]), rel=3e-4, what="dW")


# ------------------------------------------------------------------------------------------------ MMOE
@pytest.mark.parametrize("B,h,ld,K,E,T", [(24, 43, 64, 16, 3, 2), (2048, 649, 672, 128, 4, 2), (333, 100, 100, 20, 8, 4),
                                           (100, 30, 32, 300, 2, 1)])
def test_mmoe_expert_gemm_and_combine_vs_oracle(B, h, ld, K, E, T):
    """experts einsum + per-task gate softmax + gate-weighted sum (mmoe.py:86-104) and their gradients."""
    from rec_pangu_amd import functional as Fh
    g = torch.Generator().manual_seed(B + K)
    x = torch.zeros(B, ld)
    x[:, :h] = torch.randn(B, h, generator=g)
    experts = torch.rand(h, K, E, generator=g) / h ** 0.5
    ebias = torch.rand(K, E, generator=g)
    gates = [torch.randn(h, E, generator=g) * 0.3 for _ in range(T)]
    gbias = [torch.rand(E, generator=g) for _ in range(T)]
    coef = torch.randn(T, B, K, generator=g)
    # oracle: the reference's formulation
    rx, re, rb = x[:, :h].clone().requires_grad_(True), experts.clone().requires_grad_(True), ebias.clone().requires_grad_(True)
    eo = torch.einsum("ij,jkl->ikl", rx, re) + rb
    outs = [(eo * torch.softmax(rx @ gates[t] + gbias[t], dim=-1).unsqueeze(1)).sum(dim=2) for t in range(T)]
    ref = torch.stack(outs)
    (ref * coef).sum().backward()
    dx, de, db = x.to(DEV).requires_grad_(True), experts.to(DEV).requires_grad_(True), ebias.to(DEV).requires_grad_(True)
    w_cat = torch.cat([de.reshape(h, K * E)] + [t.to(DEV) for t in gates], dim=1)
    b_cat = torch.cat([db.reshape(-1)] + [t.to(DEV) for t in gbias])
    mix = torch.stack(Fh.mmoe_combine(Fh.linear_input_major(dx, w_cat, b_cat), K, E, T))  # (one [B, K] output per task)
    _close(mix.detach().cpu(), ref.detach(), what="mixtures")
    (mix * coef.to(DEV)).sum().backward()
    _close(dx.grad[:, :h].cpu(), rx.grad, rel=2e-4, what="d hidden")
    _close(de.grad.cpu(), re.grad, rel=2e-4, what="d experts")
    _close(db.grad.cpu(), rb.grad, rel=2e-4, what="d experts_bias")


# ------------------------------------------------------------------------------------------------ FM pooling
def test_fm_pool_vs_reference_fixture():
    from rec_pangu_amd.models.layers import InnerProductLayer
    ip = load_golden("layers.npz")["ip"]
    x = ip["in"].to(DEV)
    for mode in ("product_sum_pooling", "Bi_interaction_pooling"):
        from rec_pangu_amd import hip
        n0 = hip.launch_count()
        y = InnerProductLayer(output=mode)(x)
        assert hip.launch_count() > n0
        _close(y.cpu(), ip[mode], rel=1e-5, what=mode)


@pytest.mark.parametrize("B,F,D", [(1000, 26, 16), (65, 3, 4), (7, 39, 40), (300, 10, 256), (1, 1, 8)])
def test_fm_pool_fwd_bwd_vs_torch(B, F, D):
    from rec_pangu_amd import functional as Fh
    g = torch.Generator().manual_seed(B + F)
    x = torch.randn(B, F, D, generator=g)
    for bi in (False, True):
        xr = x.clone().requires_grad_(True)
        s = xr.sum(1)
        ref = 0.5 * (s * s - (xr * xr).sum(1))
        ref = ref if bi else ref.sum(-1, keepdim=True)
        w = torch.randn(ref.shape, generator=g)
        (ref * w).sum().backward()
        xd = x.to(DEV).requires_grad_(True)
        y = Fh.fm_pool(xd, bi)
        (y * w.to(DEV)).sum().backward()
        _close(y.detach().cpu(), ref.detach(), rel=2e-5, floor=1e-2, what="fm out")
        _close(xd.grad.cpu(), xr.grad, rel=2e-5, what="fm dx")


# ------------------------------------------------------------------------------------------------ BatchNorm1d
@pytest.mark.parametrize("M,N", [(4096, 64), (1000, 128), (257, 7), (2, 300), (70000, 32)])
def test_batchnorm_train_and_eval_vs_torch(M, N):
    from rec_pangu_amd import functional as Fh, hip
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g) * 3 + 5
    w = torch.randn(M, N, generator=g)
    ref_bn = torch.nn.BatchNorm1d(N)
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(N, generator=g) + 0.5)
        ref_bn.bias.copy_(torch.randn(N, generator=g))
    import copy
    bn = copy.deepcopy(ref_bn).to(DEV)
    for step in range(2):  # two steps: running statistics and num_batches_tracked follow nn.BatchNorm1d
        xr = x.clone().requires_grad_(True)
        yr = ref_bn(xr)
        (yr * w).sum().backward()
        xd = x.to(DEV).requires_grad_(True)
        n0 = hip.launch_count()
        y = Fh.batch_norm(xd, bn)
        (y * w.to(DEV)).sum().backward()
        assert hip.launch_count() > n0
        _close(y.detach().cpu(), yr.detach(), rel=2e-5, what="bn y")
        _close(xd.grad.cpu(), xr.grad, rel=1e-4, floor=1e-2, what="bn dx")
        _close(bn.weight.grad.cpu(), ref_bn.weight.grad, rel=1e-4, floor=1.0, what="bn dgamma")
        _close(bn.bias.grad.cpu(), ref_bn.bias.grad, rel=1e-4, floor=1.0, what="bn dbeta")
        _close(bn.running_mean.cpu(), ref_bn.running_mean, rel=1e-4, what="bn running_mean")
        _close(bn.running_var.cpu(), ref_bn.running_var, rel=1e-4, what="bn running_var")
        assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == step + 1
    ref_bn.eval(); bn.eval()
    xr = x.clone().requires_grad_(True)
    yr = ref_bn(xr); (yr * w).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    y = Fh.batch_norm(xd, bn); (y * w.to(DEV)).sum().backward()
    _close(y.detach().cpu(), yr.detach(), rel=2e-5, what="bn eval y")
    _close(xd.grad.cpu(), xr.grad, rel=2e-5, what="bn eval dx")


# ------------------------------------------------------------------------------------------------ EmbeddingLayer API
def test_embedding_layer_all_by_name_and_seq_vs_reference_fixture():
    """embedding.py:58-71 on the gather kernel: bit-exact rows (index work), gradients land in the named table."""
    from conftest import small_enc_dict
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import EmbeddingLayer
    g = load_golden("layers.npz")
    emb = EmbeddingLayer(small_enc_dict(), 8)
    emb.load_state_dict({k[2:]: v for k, v in g["emb"].items() if k.startswith("w/")})
    emb = emb.to(DEV)
    data = {k: v.to(DEV) for k, v in g["batch"].items()}
    n0 = hip.launch_count()
    assert torch.equal(emb(data).cpu(), g["emb"]["all"])
    one = emb(data, name="C3")
    assert one.shape == g["emb"]["by_name_C3"].shape and torch.equal(one.cpu(), g["emb"]["by_name_C3"])
    data["C3_seq"] = g["emb"]["seq_in"].to(DEV)
    seq = emb(data, name="C3_seq")
    assert seq.shape == g["emb"]["by_name_C3_seq"].shape and torch.equal(seq.cpu(), g["emb"]["by_name_C3_seq"])
    assert hip.launch_count() >= n0 + 3
    # backward of the sequence lookup: dense gradient of table C3 only, = index_add of the upstream gradient
    w = torch.randn(seq.shape, generator=torch.Generator().manual_seed(0))
    (seq * w.to(DEV)).sum().backward()
    ref = torch.zeros_like(g["emb"]["w/embedding_layer.C3.weight"])
    ref.index_add_(0, g["emb"]["seq_in"].reshape(-1), w.reshape(-1, 8))
    torch.testing.assert_close(emb.embedding_layer["C3"].weight.grad.cpu(), ref, rtol=1e-5, atol=1e-6)
    for c in ("C1", "C2", "C4", "C5"):
        gr = emb.embedding_layer[c].weight.grad
        assert gr is None or float(gr.abs().max()) == 0.0


def test_mlp_with_batchnorm_eval_vs_reference_fixture():
    """deep.py MLP(batch_norm=True, tanh/sigmoid, dropout) in eval(): Linear on MFMA, BN on rp_batchnorm_apply."""
    from rec_pangu_amd.models.layers import MLP
    g = load_golden("layers.npz")
    mlp = MLP(input_dim=43, output_dim=None, hidden_units=[16, 8], hidden_activations=["tanh", "sigmoid"],
              dropout_rates=0.1, batch_norm=True, output_activation=None)
    mlp.load_state_dict({k[2:]: v for k, v in g["mlp2"].items() if k.startswith("w/")})
    mlp = mlp.to(DEV).eval()
    y = mlp(g["mlp"]["in"].to(DEV))
    _close(y.detach().cpu(), g["mlp2"]["out"], rel=2e-5, what="mlp2 eval out")


# ------------------------------------------------------------------------------------------------ Dice (activation.py:10-34)
@pytest.mark.parametrize("case", ["a", "b"])
def test_dice_vs_reference_fixture(case):
    """rp_batchnorm_* (affine=False) + rp_dice_gate_* against the reference's Dice: training mode (batch statistics,
    running statistics after the step) and eval mode, forward and every gradient."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import Dice
    g = load_golden("dice.npz")[case]
    N = g["x"].shape[1]
    d = Dice(N)
    with torch.no_grad():
        d.alpha.copy_(g["alpha"])
    d = d.to(DEV)
    n0, tp0 = hip.launch_count(), hip.torch_path_count()
    for mode in ("train", "eval"):
        d.train(mode == "train")
        d.alpha.grad = None
        x = g["x"].to(DEV).requires_grad_(True)
        y = d(x)
        (y * g["cot"].to(DEV)).sum().backward()
        _close(y.detach().cpu(), g[f"{mode}/y"], rel=1e-5, what=f"dice {mode} y")
        _close(x.grad.cpu(), g[f"{mode}/dx"], what=f"dice {mode} dx")
        _close(d.alpha.grad.cpu(), g[f"{mode}/dalpha"], what=f"dice {mode} dalpha")
        if mode == "train":
            _close(d.bn.running_mean.cpu(), g["running_mean"], rel=1e-5, what="running_mean")
            _close(d.bn.running_var.cpu(), g["running_var"], rel=1e-5, what="running_var")
            assert int(d.bn.num_batches_tracked) == int(g["num_batches_tracked"])
    assert hip.launch_count() - n0 >= 8 and hip.torch_path_count() == tp0


def test_mlp_with_dice_activations_vs_reference_fixture():
    """deep.py:11-84 with Dice instances as hidden activations (what the reference's DIN-style towers pass): the MLP walks
    Linear -> Dice -> Linear -> Dice -> Linear on HIP kernels, no torch path."""
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import MLP, Dice
    g = load_golden("dice.npz")["mlp"]
    mlp = MLP(input_dim=10, output_dim=1, hidden_units=[16, 8], hidden_activations=[Dice(16), Dice(8)], dropout_rates=0)
    w = {k[2:]: v for k, v in g.items() if k.startswith("w/")}
    assert list(mlp.state_dict().keys()) == list(w.keys())
    mlp.load_state_dict(w)
    mlp = mlp.to(DEV).train()
    tp0 = hip.torch_path_count()
    x = g["x"].to(DEV).requires_grad_(True)
    y = mlp(x)
    (y * g["cot"].to(DEV)).sum().backward()
    assert hip.torch_path_count() == tp0
    _close(y.detach().cpu(), g["y"], what="mlp+dice y")
    _close(x.grad.cpu(), g["dx"], what="mlp+dice dx")
    for k, p in mlp.named_parameters():
        _close(p.grad.cpu(), g["g/" + k], what=f"mlp+dice grad {k}")
    for k, v in mlp.state_dict().items():
        _close(v.detach().cpu().float(), g["after/" + k].float(), rel=1e-5, what=f"mlp+dice after {k}")
