"""BASELINE configs 2/3/5 are DEFINED by the full arena: 26 Criteo tables = 33 762 603 rows x 64 fp32 = 8.64 GB, i.e. byte
offsets beyond 2^32 (row 16 777 216 onwards) and, with the gradient / moment arenas, ~35 GB live.  Every other GPU test
divides the cardinalities by 16 or 64; this module runs the hot path once at the real size, with the ids concentrated
where the address arithmetic is hardest (the top 1 % of the three 7-10 M-row tables, the OOV row of every table), and
checks size-independent properties — there is no CPU oracle run at this size:

  * index work bit-exact: arena-row keys == base[f] + id, every gathered row == arena[key] (plain and fused gather),
    radix sort == torch.sort(stable=True) (keys AND positions);
  * rp_embed_grad_gemm is linear: column sums of the gradient arena == column sums of the per-pair gradients, and its
    touched-row set is exactly the looked-up set;
  * 3 lazy-Adam train steps of the whole DeepFM == the dense kernel run on the same gradients, bit for bit on every row
    of the arena (touched, replayed and never-touched alike), moments included.
(reference: rec_pangu/models/layers/embedding.py:58-63, trainer.py:75)"""
import os
import sys

import pytest
import torch

from conftest import require_gpu

pytestmark = pytest.mark.gpu
DEV = "cuda"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module")
def hip():
    require_gpu()
    from rec_pangu_amd import hip as h
    h.lib()
    if torch.cuda.get_device_properties(0).total_memory < 120 * 2 ** 30:
        pytest.fail("the full-arena test needs the 288 GB of an MI355X")
    return h


def _hard_ids(rows, B, gen):
    """per table: first half uniform over the table, second half from its top 1 % (byte offsets > 2^32 for the big tables
    behind row 16.7 M), plus the last row (the OOV id = vocab_size) and row 0 pinned in."""
    idx = []
    for r in rows:
        lo = torch.randint(0, r, (B // 2,), generator=gen, device=DEV)
        top = r - 1 - torch.randint(0, max(1, r // 100), (B - B // 2,), generator=gen, device=DEV)
        t = torch.cat([lo, top])
        t[0], t[1] = r - 1, 0
        idx.append(t[torch.randperm(B, generator=gen, device=DEV)].contiguous())
    return idx


def test_full_criteo_arena_index_work_and_gradient_linearity(hip):
    import bench
    rows = [c + 1 for c in bench.CRITEO_CARD]
    F, D, B, ND, H = len(rows), 64, 65536, 13, 64
    total = sum(rows)
    assert total == 33762603 and total * D * 4 > 2 ** 33
    gen = torch.Generator(device=DEV).manual_seed(7)
    arena = torch.randn(total, D, generator=gen, device=DEV)
    base_l = [0]
    for r in rows[:-1]:
        base_l.append(base_l[-1] + r)
    base = torch.tensor(base_l, dtype=torch.int64, device=DEV)
    cnt = torch.tensor(rows, dtype=torch.int64, device=DEV)
    idx = _hard_ids(rows, B, gen)
    exp_keys = torch.cat([base_l[f] + idx[f] for f in range(F)])
    assert int((exp_keys * D * 4 >= 2 ** 32).sum()) > B, "the ids must reach past the 4 GB byte offset"
    dense = [torch.rand(B, generator=gen, device=DEV) for _ in range(ND)]
    K = F * D + ND
    ldx = (K + 63) // 64 * 64
    err = torch.zeros(1, dtype=torch.int32, device=DEV)

    # ---- plain gather + keys
    x, fm, ssum, keys = hip.embed_gather_fwd(arena, base, cnt, idx, dense, ldx, True, True, True, err)
    assert int(err.item()) == 0
    assert torch.equal(keys.long(), exp_keys), "arena-row keys must be bit-exact"
    emb = x[:, :F * D].view(B, F, D)
    for f in range(F):
        assert torch.equal(emb[:, f], arena[exp_keys[f * B:(f + 1) * B]]), f"field {f}: gathered rows differ"
    assert torch.equal(x[:, F * D:K], torch.stack(dense, dim=1))
    torch.testing.assert_close(ssum, emb.sum(1), rtol=1e-5, atol=1e-5)
    # rp_embed_keys (what the lazy optimizer / sort-ahead path uses)
    k2 = hip.embed_keys(base, cnt, idx, err)
    assert torch.equal(k2, keys)
    # an id one past a big table's OOV row is flagged
    bad = [t.clone() for t in idx]
    bad[2][5] = rows[2]
    hip.embed_keys(base, cnt, bad, err)
    assert int(err.item()) != 0
    err.zero_()

    # ---- fused gather + first Linear: same rows, same keys
    W = (torch.randn(H, K, generator=gen, device=DEV) / K ** 0.5)
    Wp = torch.zeros(H, (K + 3) // 4 * 4, device=DEV)
    Wp[:, :K] = W
    Wd = Wp[:, :K]
    bias = torch.randn(H, generator=gen, device=DEV) * 0.1
    hip.set_matmul_precision("bf16x6")
    try:
        assert hip.embed_gather_linear_fits(D, F, ND, H, ldx, Wd)
        x1, h1, fm1, s1, k1 = hip.embed_gather_linear_fwd(arena, base, cnt, idx, dense, ldx, Wd, bias, True, True, True, err)
        assert int(err.item()) == 0
        assert torch.equal(x1[:, :K], x[:, :K]) and torch.equal(k1, keys)
        pre = x[:, :K].double() @ W.double().T + bias.double()
        scale = float(pre.abs().max())
        assert float((h1.double() - pre.clamp_min(0)).abs().max()) <= 4e-5 * scale
        del x1, h1, pre

        # ---- sort: keys and positions == torch.sort(stable=True)
        end_bit = max(1, (total - 1).bit_length())
        sk, sp = hip.sort_pairs(keys, end_bit=end_bit)
        rk, rp_ = torch.sort(keys, stable=True)
        assert torch.equal(sk, rk) and torch.equal(sp.long(), rp_), "radix sort differs from torch.sort(stable=True)"

        # ---- rp_embed_grad_gemm: linearity + touched set
        dh = torch.randn(B, H, generator=gen, device=DEV) * 1e-3
        gfm = torch.randn(B, 1, generator=gen, device=DEV) * 1e-3
        wt = hip.transpose(Wd, rows_out=ldx)
        assert hip.embed_grad_gemm_fits(D, H, dh, wt)
        G = torch.zeros_like(arena)
        hip.embed_grad_gemm(sk, sp, B, D, dh, wt, None, gfm, ssum, arena, G, accumulate=False)
        G2 = torch.zeros_like(arena)
        hip.embed_grad_gemm(sk, sp, B, D, dh, wt, None, gfm, ssum, arena, G2, accumulate=False)
        assert torch.equal(G, G2), "two launches of the fused gather backward differ"
        del G2
        dX = (dh.double() @ W.double()[:, :F * D]).view(B, F, D)
        dX = dX + gfm.double()[:, :, None] * (ssum.double()[:, None, :] - emb.double())
        col = dX.sum(dim=(0, 1))
        got = G.double().sum(0)
        assert float((got - col).abs().max()) <= 1e-4 * float(col.abs().max() + dX.abs().max() * 50)
        touched = torch.zeros(total, dtype=torch.bool, device=DEV)
        touched[exp_keys] = True
        nz = (G != 0).any(dim=1)
        assert not bool((nz & ~touched).any()), "a row nobody looked up received a gradient"
        # spot rows: the last row of the arena (largest byte offset) and a hot tiny-table row, against an fp64 sum
        for row in (total - 1, base_l[8] + 1, int(exp_keys[B * 2 + 17])):
            pairs = (exp_keys == row).nonzero().flatten()
            ref = dX.view(B * F, D)[(pairs % B) * F + pairs // B].sum(0)
            assert float((G[row].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max() + 1e-6) * max(1, pairs.numel()) ** 0.5
    finally:
        hip.set_matmul_precision("auto")


def test_full_criteo_arena_three_lazy_adam_steps_equal_the_dense_kernel(hip):
    import bench
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import FusedAdam
    enc = bench.criteo_enc_dict()
    rows = [c + 1 for c in bench.CRITEO_CARD]
    F, B = len(rows), 65536
    torch.manual_seed(0)
    with torch.device(DEV):
        model = DeepFM(embedding_dim=64, hidden_units=[64, 64, 64], enc_dict=enc)
    model.train()
    emb = model.embedding_layer
    assert emb.arena.shape[0] == 33762603
    opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=False, lazy_tables=True)
    pd = emb.arena.detach().clone()
    md, vd = torch.zeros_like(pd), torch.zeros_like(pd)
    gen = torch.Generator(device=DEV).manual_seed(11)
    n0 = hip.launch_count()
    for t in range(1, 4):
        ids = _hard_ids(rows, B, gen)
        batch = {f"I{i + 1}": torch.rand(B, generator=gen, device=DEV) for i in range(13)}
        batch.update({f"C{i + 1}": ids[i] for i in range(F)})
        batch["label"] = (torch.rand(B, generator=gen, device=DEV) < 0.25).float()
        out = model(batch)
        assert bool(torch.isfinite(out["loss"]))
        out["loss"].backward()
        assert emb.grads_are_arena()
        gd = emb.grad_arena.clone()
        hip.adam_step([pd.view(-1)], [gd.view(-1)], [md.view(-1)], [vd.view(-1)], 1e-3, 0.9, 0.999, 1e-8, t, zero_grad=False)
        del gd
        opt.step()
        model.zero_grad()
        # the rows this step touched are current: equal to the dense state already
        keys = torch.cat([emb.row_base[f] + ids[f] for f in range(F)])
        assert torch.equal(emb.arena.detach()[keys], pd[keys]), f"step {t}: touched rows differ from the dense kernel"
    assert hip.launch_count() > n0
    lz = emb._lazy
    assert lz is not None and lz.t == 3
    behind = int(((lz.last > 0) & (lz.last < 3)).sum())
    assert behind > 100000, "rows touched in an earlier step only must be waiting for their replay"
    opt.flush()
    assert torch.equal(emb.arena.detach(), pd), "lazy Adam differs from the dense kernel somewhere in the 8.6 GB arena"
    assert torch.equal(lz.m, md) and torch.equal(lz.v, vd)
    never = (lz.last == 0)
    assert int(never.sum()) > 20_000_000
    # rows nobody ever looked up have zero moments and have not moved
    assert not bool(md[never].any()) and not bool(vd[never].any())
    del model, opt, pd, md, vd
    torch.cuda.empty_cache()
