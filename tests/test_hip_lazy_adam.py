"""Exact lazy dense Adam (rp_lazy_adam_rows / rp_lazy_adam_flush) against the dense kernel (rp_adam_step):
the reference's optimiser is DENSE Adam over every embedding row (trainer.py:75); the lazy execution
must give bit-identical parameters and moments, including learning-rate changes between steps, rows touched
several times, rows never touched, and mid-run replays."""
import pytest
import torch

from conftest import load_golden, require_gpu
from test_host_models import CASES, build

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    from rec_pangu_amd import hip
    hip.lib()


# 1: the LR_Layer's tables (scalar rows); 6: not a multiple of 4; 64 / 40 / 100 / 200: the one-row-per-wave replay with
# 1, 1 (partly filled), 2 and 4 floats per lane
# steps = 97: replay chains of up to ~96 skipped steps (odd and even lengths: the replay loop is unrolled by two)
@pytest.mark.parametrize("D,steps", [(64, 14), (1, 14), (6, 14), (40, 14), (100, 14), (200, 14), (64, 97), (40, 97),
                                     (1, 97)])
def test_lazy_rows_bit_identical_to_dense_kernel(D, steps):
    from rec_pangu_amd import hip
    g = torch.Generator().manual_seed(0)
    R = 5000
    p0 = torch.randn(R, D, generator=g)
    b1, b2, eps = 0.9, 0.999, 1e-8
    pd, md, vd = p0.to(DEV), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV)
    pl, ml, vl = p0.to(DEV), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV)
    last = torch.zeros(R, dtype=torch.int32, device=DEV)
    table = torch.zeros(steps + 1, 2)
    hot = torch.arange(0, 40)  # rows touched (almost) every step; the rest rarely or never
    for t in range(1, steps + 1):
        lr = 1e-2 * (0.7 ** (t // 4))  # a scheduler changing lr mid-run
        table[t] = torch.tensor(hip.adam_step_scalars(lr, b1, b2, t))
        dev_table = table.to(DEV)
        n = 300
        # (long runs: after step 10 only the first quarter is touched, so the rows of the second quarter that were
        # updated early carry replay chains of up to steps - 1 skipped steps into the peeks and the final flush)
        hi = R // 2 if (t <= 10 or steps <= 14) else R // 4
        rows = torch.cat([hot[torch.rand(40, generator=g) < 0.9], torch.randint(40, hi, (n,), generator=g)])
        rows = rows[torch.randperm(rows.numel(), generator=g)]
        grad_rows = torch.randn(rows.numel(), D, generator=g)
        gd = torch.zeros(R, D).index_add_(0, rows, grad_rows).to(DEV)
        gl = gd.clone()
        # a forward of some other rows replays them first (must not change the final result)
        peek = torch.randint(0, R, (200,), generator=g).to(torch.int32).to(DEV)
        sk, _ = hip.sort_pairs(peek, end_bit=13)
        hip.lazy_adam_rows(sk, D, pl, None, ml, vl, last, dev_table, t - 1, False, False, b1, b2, eps)
        # dense reference step
        hip.adam_step([pd.view(-1)], [gd.view(-1)], [md.view(-1)], [vd.view(-1)], lr, b1, b2, eps, t, zero_grad=True)
        # lazy step over the touched rows only (sorted, with duplicates)
        sk, _ = hip.sort_pairs(rows.to(torch.int32).to(DEV), end_bit=13)
        hip.lazy_adam_rows(sk, D, pl, gl, ml, vl, last, dev_table, t, True, True, b1, b2, eps)
        assert torch.count_nonzero(gl) == 0, "the lazy step must clear the gradient rows it consumed"
        # rows that are current (last == t) already equal the dense state bit for bit
        cur = (last == t).nonzero().flatten()
        assert cur.numel() > 0
        assert torch.equal(pl[cur], pd[cur]) and torch.equal(ml[cur], md[cur]) and torch.equal(vl[cur], vd[cur])
    assert int((last == 0).sum()) > R // 3, "rows never touched stay at last == 0"
    hip.lazy_adam_flush(R, D, pl, ml, vl, last, dev_table, steps, b1, b2, eps)
    assert torch.equal(pl, pd), (pl - pd).abs().max()
    assert torch.equal(ml, md) and torch.equal(vl, vd)
    never = (last == 0).nonzero().flatten()
    assert torch.equal(pl[never].cpu(), p0[never.cpu()]), "never-updated rows must be untouched"


@pytest.mark.parametrize("name", ["deepfm", "xdeepfm", "dcn"])
def test_model_lazy_equals_dense_and_reference(name):
    """Whole model, 4 steps on alternating batches: lazy and dense FusedAdam give bit-identical weights, and
    after 2 steps on the golden batch both match the reference's own Adam run."""
    from rec_pangu_amd.optim import FusedAdam
    g = load_golden(f"model_{name}.npz")
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    other = {k: (v.flip(0) if v.dtype.is_floating_point else torch.zeros_like(v)) for k, v in batch.items()}
    finals = {}
    for lazy in (False, True):
        model = build(name).to(DEV)
        model.train(CASES[name][1])
        opt = FusedAdam(model.parameters(), lr=1e-2, fuse_zero_grad=True, lazy_tables=lazy)
        for i in range(2):
            model(batch)["loss"].backward()
            opt.step()
            model.zero_grad()
        sd2 = {k: v.clone() for k, v in model.state_dict().items()}  # state_dict() flushes the lazy rows
        for k, v in g["adam2"].items():
            if v.dtype.is_floating_point:
                tol = 2e-4 * max(1e-2, float(v.abs().max()))
                assert (sd2[k].cpu() - v).abs().max() <= tol, f"{name} lazy={lazy}: {k}"
        for i in range(2):
            model(other)["loss"].backward()
            opt.step()
            model.zero_grad()
        model.eval()
        with torch.no_grad():
            pred = model(batch, is_training=False)["pred"]  # looks rows up again: replays them
        finals[lazy] = ({k: v.clone() for k, v in model.state_dict().items()}, pred.clone(),
                        {k: (s["exp_avg"].clone(), s["exp_avg_sq"].clone()) for k, s in
                         zip(range(len(opt.state)), opt.state_dict()["state"].values())})
    for k in finals[False][0]:
        assert torch.equal(finals[False][0][k], finals[True][0][k]), f"{name}: {k} differs between dense and lazy"
    assert torch.equal(finals[False][1], finals[True][1])
    for k in finals[False][2]:
        assert torch.equal(finals[False][2][k][0], finals[True][2][k][0]), f"exp_avg {k}"
        assert torch.equal(finals[False][2][k][1], finals[True][2][k][1]), f"exp_avg_sq {k}"


def test_scalar_table_growth_keeps_lazy_exact(monkeypatch):
    """The per-step scalar table grows in chunks (LazyAdamRows.TABLE_CHUNK = 1024 steps): with a chunk of 3 a 10-step
    run with a learning-rate change crosses several extensions and must still equal the dense optimizer bit for bit."""
    from rec_pangu_amd.optim import FusedAdam, LazyAdamRows
    monkeypatch.setattr(LazyAdamRows, "TABLE_CHUNK", 3)
    g = load_golden("model_deepfm.npz")
    batch = {k: v.to(DEV) for k, v in g["batch"].items()}
    other = {k: (v.flip(0) if v.dtype.is_floating_point else torch.zeros_like(v)) for k, v in batch.items()}
    finals = {}
    for lazy in (False, True):
        model = build("deepfm").to(DEV)
        model.train(CASES["deepfm"][1])
        opt = FusedAdam(model.parameters(), lr=1e-2, fuse_zero_grad=True, lazy_tables=lazy)
        for i in range(10):
            if i == 6:
                for grp in opt.param_groups:
                    grp["lr"] = 3e-3
            model(batch if i % 3 else other)["loss"].backward()
            opt.step()
            model.zero_grad()
        finals[lazy] = {k: v.clone() for k, v in model.state_dict().items()}
    for k in finals[False]:
        assert torch.equal(finals[False][k], finals[True][k]), k
