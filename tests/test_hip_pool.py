"""GPU parity of the pooled multi-id lookup (north_star: "fused CSR/segmented embedding gather + sum-pool"):
rp_embed_gather_pool_fwd / rp_embed_pool_bwd / rp_seq_pool_* against the reference's `_seq` lookup followed by
MaskedSumPooling / MaskedAveragePooling (embedding.py:64-71, layers/sequence.py:13-59; tests/golden/pool.npz) and against the
CPU oracle on larger seeded inputs: ragged CSR bags, empty bags, a padding id that is in every bag (one run of half the
pairs in the backward), rows wider than a wave, the lazy optimizer."""
import pytest
import torch

from conftest import load_golden, require_gpu
from oracle import ref_ops as R
from test_oracle_golden import pool_case

pytestmark = pytest.mark.gpu
DEV = "cuda"
ENC = {"C1": {"vocab_size": 7}, "I1": {"min": 0.0, "max": 1.0}, "hist": {"vocab_size": 60}, "C2": {"vocab_size": 3}}


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


def _layer(enc, D, table=None, name="hist"):
    from rec_pangu_amd.models.layers import EmbeddingLayer
    emb = EmbeddingLayer(enc, D)
    if table is not None:
        with torch.no_grad():
            emb.embedding_layer[name].weight.copy_(table)
    return emb.to(DEV)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_lookup_pooled_vs_reference_fixture(case):
    from rec_pangu_amd import hip
    from rec_pangu_amd.models.layers import MaskedAveragePooling, MaskedSumPooling
    table, seq, modes = pool_case(load_golden("pool.npz"), case)
    emb = _layer(ENC, table.shape[1], table)
    X = {"hist_seq": seq.to(DEV)}
    for mode, (out, cot, grad) in modes.items():
        pooling = "sum" if mode == "sum" else "average"
        emb.zero_grad()
        n0 = hip.launch_count()
        y = emb.lookup_pooled(X, "hist_seq", pooling)
        assert hip.launch_count() == n0 + 1, "lookup + pooling must be ONE launch"
        torch.testing.assert_close(y.cpu(), out, rtol=1e-6, atol=1e-7)
        (y * cot.to(DEV)).sum().backward()
        g = emb.embedding_layer["hist"].weight.grad.cpu()
        # (empty bags of the masked average hand 1e16 * cot to the padding row: compare relative to the row scale)
        torch.testing.assert_close(g, grad, rtol=2e-6, atol=1e-6 * float(grad.abs().max()))
        for c in ("C1", "C2"):
            gr = emb.embedding_layer[c].weight.grad
            assert gr is None or float(gr.abs().max()) == 0.0
        # the unfused composition on the device: `_seq` lookup kernel, then the pooling module's own kernel
        emb.zero_grad()
        mod = MaskedSumPooling() if mode == "sum" else MaskedAveragePooling()
        y2 = mod(emb(X, name="hist_seq"))
        torch.testing.assert_close(y2.cpu(), out, rtol=1e-6, atol=1e-7)
        (y2 * cot.to(DEV)).sum().backward()
        torch.testing.assert_close(emb.embedding_layer["hist"].weight.grad.cpu(), grad, rtol=2e-6,
                                   atol=1e-6 * float(grad.abs().max()))
        if case != "a":  # CSR bags without the all-zero padding ids: same sums and averages, same table gradient except row 0
            keep = seq != 0
            offsets = torch.cat([torch.zeros(1, dtype=torch.long), keep.sum(1).cumsum(0)])
            emb.zero_grad()
            y3 = emb.lookup_pooled({"hist_seq": seq[keep].to(DEV)}, "hist_seq", pooling, offsets=offsets.to(DEV))
            torch.testing.assert_close(y3.cpu(), out, rtol=1e-6, atol=1e-7)
            (y3 * cot.to(DEV)).sum().backward()
            g3 = emb.embedding_layer["hist"].weight.grad.cpu()
            torch.testing.assert_close(g3[1:], grad[1:], rtol=2e-6, atol=1e-6 * float(grad[1:].abs().max()))
            assert float(g3[0].abs().max()) == 0.0


@pytest.mark.parametrize("D,V,B,Lmax", [(64, 200000, 8192, 50), (16, 5000, 4096, 20), (200, 3000, 1000, 7), (6, 50, 300, 33)])
def test_pooled_csr_vs_oracle(D, V, B, Lmax):
    """ragged CSR bags (empty ones included) and the dense form with a padding id 0 in every bag, against the oracle:
    forward 1e-6, table gradient 1e-5 of its scale; the gradient is bit-identical from run to run (no fp atomics)."""
    gen = torch.Generator().manual_seed(D + B)
    enc = {"a": {"vocab_size": 10}, "hist": {"vocab_size": V}}
    torch.manual_seed(0)
    emb = _layer(enc, D)
    table = emb.embedding_layer["hist"].weight.detach().cpu().clone()
    lens = torch.randint(0, Lmax + 1, (B,), generator=gen)
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    ids = torch.randint(0, V + 1, (int(offsets[-1]),), generator=gen)
    ids[torch.rand(ids.numel(), generator=gen) < 0.2] = 3  # a hot row
    cot = torch.randn(B, D, generator=gen)
    bag = torch.repeat_interleave(torch.arange(B), lens)
    for pooling in ("sum", "average"):
        # float64 restatement of R.embedding_bags_pooled (checked against it on the first 64 bags): the bag-by-bag oracle
        # is a python loop, and fp32 sums of up to 50 rows / 40 k gradient rows carry their own rounding
        w = table.double().requires_grad_(True)
        rows = w[ids]
        ref = torch.zeros(B, D, dtype=torch.float64).index_add(0, bag, rows)
        if pooling == "average":
            cnt = torch.zeros(B, D, dtype=torch.float64).index_add(0, bag, (rows.detach().float() != 0).double())
            ref = ref / (cnt + 1e-16)
        small = R.embedding_bags_pooled(table, ids[:int(offsets[64])], offsets[:65], pooling)
        torch.testing.assert_close(ref[:64].detach().float(), small, rtol=1e-5, atol=1e-5)
        (ref * cot.double()).sum().backward()
        grads = []
        for rep in range(2):
            emb.zero_grad()
            y = emb.lookup_pooled({"hist_seq": ids.to(DEV)}, "hist_seq", pooling, offsets=offsets.to(DEV))
            (y * cot.to(DEV)).sum().backward()
            grads.append(emb.embedding_layer["hist"].weight.grad.clone())
        assert torch.equal(grads[0], grads[1]), "the pooled backward must be deterministic"
        live = lens > 0  # (an empty bag of the masked average is 0 / 1e-16 = 0 on both sides)
        err = float((y.detach().cpu().double() - ref.detach()).abs().max())
        tol = 1e-5 * max(1.0, float(ref.detach()[live].abs().max()))
        gerr = float((grads[0].cpu().double() - w.grad).abs().max())
        # (the hot row sums 20 % of all gradient rows — 8e4 of them in the largest case — in fp32: ~1e-4 of its value)
        gtol = 2e-4 * float(w.grad.abs().max())
        cold = torch.ones(V + 1, dtype=torch.bool)
        cold[3] = False
        assert float((grads[0].cpu().double() - w.grad)[cold].abs().max()) <= 1e-5 * float(w.grad[cold].abs().max())
        print(f"\n{pooling} D={D} nnz={ids.numel()}: fwd err {err:.2e} (tol {tol:.2e}), grad err {gerr:.2e} (tol {gtol:.2e})")
        assert err <= tol and gerr <= gtol, pooling
    # dense [B, L] bags padded with id 0: half of all pairs carry the same key
    L = Lmax
    seq = torch.randint(1, V + 1, (B, L), generator=gen) * (torch.arange(L)[None, :] < lens.clamp(max=L)[:, None])
    w = table.clone().requires_grad_(True)
    ref = R.embedding_seq_pooled({"hist": w}, {"hist_seq": seq}, "hist_seq", "sum")
    (ref * cot).sum().backward()
    emb.zero_grad()
    y = emb.lookup_pooled({"hist_seq": seq.to(DEV)}, "hist_seq", "sum")
    (y * cot.to(DEV)).sum().backward()
    assert float((y.detach().cpu() - ref.detach()).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()))
    g = emb.embedding_layer["hist"].weight.grad.cpu()
    assert float((g - w.grad).abs().max()) <= 2e-4 * float(w.grad.abs().max())  # (row 0: half of all pairs, fp32 on both sides)


def test_pooled_lookup_out_of_range_id_raises():
    emb = _layer(ENC, 8)
    seq = torch.randint(0, 61, (16, 4))
    seq[3, 2] = 61  # vocab 60 -> rows 0..60
    with pytest.raises(IndexError):
        emb.lookup_pooled({"hist_seq": seq.to(DEV)}, "hist_seq", "sum")


def test_pooled_lookup_trains_with_lazy_adam():
    """a pooled `_seq` feature next to the ordinary fields, FusedAdam with lazy tables (closed-form and serial replay)
    against dense execution: the pooled lookup replays the rows it reads before reading them."""
    from rec_pangu_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(2)
    enc = {"u": {"vocab_size": 50}, "hist": {"vocab_size": 3000}}
    B, L, D = 128, 6, 16
    batches = [{"u": torch.randint(0, 51, (B,), generator=gen).to(DEV),
                "hist_seq": torch.randint(0, 3001, (B, L), generator=gen).to(DEV),
                "y": torch.rand(B, D, generator=gen).to(DEV)} for _ in range(6)]
    finals = {}
    for kind in ("dense", "lazy"):
        torch.manual_seed(1)
        emb = _layer(enc, D)
        opt = FusedAdam(emb.parameters(), lr=1e-2, fuse_zero_grad=True, lazy_tables=(kind == "lazy"))
        for i in range(24):
            b = batches[i % 6]
            pooled = emb.lookup_pooled(b, "hist_seq", "average" if i % 2 else "sum")
            one = emb(b, name="u").squeeze(1)
            ((pooled + one - b["y"]) ** 2).mean().backward()
            opt.step()
            emb.zero_grad()
        finals[kind] = {k: v.clone() for k, v in emb.state_dict().items()}
    for k in finals["dense"]:
        assert torch.equal(finals["dense"][k], finals["lazy"][k]), k
