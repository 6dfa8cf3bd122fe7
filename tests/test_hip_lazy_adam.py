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
    from rec_pangu_amd.optim import FusedAdam, LazyAdamRows, StepTables
    monkeypatch.setattr(LazyAdamRows, "TABLE_CHUNK", 3)
    monkeypatch.setattr(StepTables, "MIN_CAPACITY", 0)  # (so that the buffers themselves double on the way)
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


def _adam_tables(steps, lr_of, b1, b2, eps):
    """host tables by step: float {A_t, B_t} (the kernels' scalars) and double {-lr_t/(1-b1^t), 1/sqrt(1-b2^t)}"""
    from rec_pangu_amd import hip
    table = torch.zeros(steps + 1, 2)
    ns_d = torch.zeros(steps + 1, 2, dtype=torch.float64)
    for t in range(1, steps + 1):
        table[t] = torch.tensor(hip.adam_step_scalars(lr_of(t), b1, b2, t, eps))
        ns_d[t, 0] = -lr_of(t) / (1.0 - b1 ** t)
        ns_d[t, 1] = 1.0 / (1.0 - b2 ** t) ** 0.5
    return table, ns_d


# closed-form replay (rp_lazy_adam_cf_table + cf_table argument of rows / flush): the skipped zero-gradient steps after
# step 256 are evaluated in one go.  Two sides are run on the same touches — the serial (bit-exact) replay and the closed
# form — and BOTH are compared with a float64 dense Adam on the host, the arithmetic both approximate:
#   * the serial replay rounds p, m and s once per skipped step: a random walk of ~sqrt(k) half-ulps (k up to ~900
#     here: ~1e-6 rms, a few 1e-6 at the maximum over 2e5 elements); the closed form rounds a handful of times per
#     replay and truncates its series at 9e-8 — so "closed == serial to 1e-6" is not a meaningful bar, the serial side
#     itself is not that close to exact arithmetic;
#   * the gate: per element, the closed form is not further from float64 than the serial replay (x1.5 + 1e-7), and the
#     two agree within the sum of their float64 errors (reported, bounded by 1e-5 of the parameter scale).
@pytest.mark.parametrize("D", [64, 40, 1])
def test_closed_form_replay_vs_serial_and_float64(D):
    from rec_pangu_amd import hip
    g = torch.Generator().manual_seed(0)
    R, steps, CF_FROM = 3000, 1200, 256
    b1, b2, eps = 0.9, 0.999, 1e-8
    lr_of = lambda t: 1e-3 * (0.5 ** (t // 500))  # noqa: E731  (a scheduler stepping mid-run)
    table, ns_d = _adam_tables(steps, lr_of, b1, b2, eps)
    dev_table, dev_nsd = table.to(DEV), ns_d.to(DEV)
    cf = torch.zeros(steps + 8, 8, device=DEV)
    p0 = 0.05 * torch.randn(R, D, generator=g)
    st = {k: [p0.to(DEV), torch.zeros(R, D, device=DEV), torch.zeros(R, D, device=DEV),
              torch.zeros(R, dtype=torch.int32, device=DEV)] for k in ("serial", "closed")}
    p64, m64, v64 = p0.double(), torch.zeros(R, D, dtype=torch.float64), torch.zeros(R, D, dtype=torch.float64)
    hot = torch.arange(0, 20)

    built = {"t": -1}  # the table is brought forward step by step (only the stamps that are not final yet are rebuilt)

    def cf_args(kind, t_end):
        if kind == "serial" or t_end <= CF_FROM:
            return None, 0
        if built["t"] != t_end:
            hip.lazy_adam_cf_table(dev_nsd, t_end, CF_FROM, b1, b2, cf, built_to=built["t"])
            built["t"] = t_end
        return cf, CF_FROM

    for t in range(1, steps + 1):
        # ~6 cold rows per step out of 1500: revisit gaps are geometric with mean 250; the
        # gradient scale differs per row and per column by orders of magnitude (eps/s from ~1e-5 to >1: both ends of the
        # expansion variable)
        rows = torch.cat([hot, torch.randint(20, R // 2, (6,), generator=g)])
        if t <= 330:  # rows of the third quarter are only touched early: the final flush owes them up to ~900 steps
            rows = torch.cat([rows, torch.randint(R // 2, 3 * R // 4, (8,), generator=g)])
        scale = 10.0 ** (-1 - 5 * torch.rand(rows.numel(), 1, generator=g)) * 10.0 ** (-2 * torch.rand(1, D, generator=g))
        grad_rows = torch.randn(rows.numel(), D, generator=g) * scale
        gd = torch.zeros(R, D).index_add_(0, rows, grad_rows)
        peek = torch.randint(0, R // 2, (40,), generator=g).to(torch.int32).to(DEV)  # (never the third quarter)
        skp, _ = hip.sort_pairs(peek, end_bit=13)
        sk, _ = hip.sort_pairs(rows.to(torch.int32).to(DEV), end_bit=13)
        for kind, (p, m, v, last) in st.items():
            c, cfrom = cf_args(kind, t - 1)
            if t > 1:
                hip.lazy_adam_rows(skp, D, p, None, m, v, last, dev_table, t - 1, False, False, b1, b2, eps, c, cfrom)
            hip.lazy_adam_rows(sk, D, p, gd.to(DEV), m, v, last, dev_table, t, True, True, b1, b2, eps, c, cfrom)
        # float64 dense Adam (torch.optim.Adam's formulas)
        m64.mul_(b1).add_(gd.double(), alpha=1 - b1)
        v64.mul_(b2).addcmul_(gd.double(), gd.double(), value=1 - b2)
        p64.addcdiv_(m64, v64.sqrt() / (1 - b2 ** t) ** 0.5 + eps, value=-lr_of(t) / (1 - b1 ** t))
    owed = steps - st["closed"][3].cpu()[st["closed"][3].cpu() > 0]
    assert int(owed.max()) > 600, "the final flush must see long replay chains"
    for kind, (p, m, v, last) in st.items():
        c, cfrom = cf_args(kind, steps)
        hip.lazy_adam_flush(R, D, p, m, v, last, dev_table, steps, b1, b2, eps, c, cfrom)
    ps, ms, vs, _ = [x.cpu() for x in st["serial"]]
    pc, mc, vc, lastc = [x.cpu() for x in st["closed"]]
    assert torch.equal(lastc > 0, st["serial"][3].cpu() > 0)
    never = lastc == 0
    assert torch.equal(pc[never], p0[never]), "never-updated rows must be untouched"
    pscale = torch.maximum(ps.abs(), torch.tensor(1e-2))
    row_m = ms.abs().amax(1, keepdim=True).clamp_min(1e-30)
    row_v = vs.abs().amax(1, keepdim=True).clamp_min(1e-30)
    s64 = v64.sqrt()
    long_rows = torch.zeros(R, dtype=torch.bool)
    long_rows[R // 2:3 * R // 4] = True
    long_rows &= lastc > 0  # touched before step 330, never since: ONE replay of 870+ steps at the flush
    assert int(long_rows.sum()) > 200

    def errs(a, ref, scale, rows):
        e = ((a.double() - ref).abs() / scale)[rows]
        return float(e.max()), float(e.pow(2).mean().sqrt())

    for what, rows in (("all rows", lastc > 0), ("rows with one 870+-step replay", long_rows)):
        print(f"\nD={D}, {what}: (max, rms) error against float64, relative to the parameter / row scale")
        for name, ser, clo, ref, scale in (("p", ps, pc, p64, pscale), ("m", ms, mc, m64, row_m), ("s", vs, vc, s64, row_v)):
            (xs, rs), (xc, rc) = errs(ser, ref, scale, rows), errs(clo, ref, scale, rows)
            d = float(((clo - ser).abs() / scale)[rows].max())
            print(f"   {name}: serial {xs:.2e} {rs:.2e} | closed {xc:.2e} {rc:.2e} | closed vs serial max {d:.2e}")
            # the closed form is not further from float64 than the serial fp32 replay (max statistics over ~1e5 elements
            # are noisy: x3; rms: x1.25), and the two agree to 1e-5 of the scale
            assert xc <= 3 * xs + 1e-7 and rc <= 1.25 * rs + 1e-8, (what, name)
            assert d <= 1e-5, (what, name, d)


def test_closed_form_replay_model_level():
    """DeepFM at batch 256 over tables of 20 k - 50 k rows (rows are behind by ~100-1000 steps when they are read).

    (1) 1000 training steps with the serial replay, then the SAME state is read through both replays: a held-out batch
        evaluated by the model (serial) and by its deep copy switched to the closed form — logits within 1e-5
        (north_star: 1e-4; measured 3e-7), the replayed rows within 1e-6 of the weight scale (measured 4e-7).
    (2) both modes trained in lockstep from the same initial weights: through step 700 (444 steps after the closed form
        takes over at step 256) predictions agree to 1e-5 (measured 7e-7) and the dense parameters to 1e-6 (measured
        2e-7).

    Why (2) stops at 700 and is not "1e-4 after 1100 steps": Adam divides by sqrt(v), so a dense parameter whose gradient
    hovers around zero (a ReLU unit that is almost never active) takes +-lr steps whose SIGN depends on the last bits of
    that gradient.  On this seed such an event separates the two runs between steps 701 and 751 (dense parameters: 1.8e-7
    -> 2.9e-4 apart within 50 steps, predictions 1e-2 apart at step 1100); from then on the two are different, equally
    valid fp32 trajectories of the same optimizer (profiles/microbench/probes/diag_cf2.py prints the trace).  The serial replay is no
    reference point in that comparison: against float64 Adam the closed form is the MORE accurate of the two
    (test_closed_form_replay_vs_serial_and_float64; profiles/microbench/probes/diag_cf.py: 1.5e-6 rms of the update against 2e-5).  The
    final divergence is printed, not asserted."""
    import copy
    from rec_pangu_amd.models.ranking import DeepFM
    from rec_pangu_amd.optim import FusedAdam
    enc = {f"I{i}": {"min": 0.0, "max": 1.0} for i in range(4)}
    enc.update({f"C{i}": {"vocab_size": v} for i, v in enumerate([20000, 30, 7000, 3, 50000, 900])})
    gen = torch.Generator().manual_seed(3)
    B, NB = 256, 1101  # a fresh batch every step: revisit gaps are geometric with means 78 / 195 steps at 20 k / 50 k rows

    big = {f"I{i}": torch.rand(NB, B, generator=gen).to(DEV) for i in range(4)}
    big.update({f"C{i}": torch.randint(0, enc[f"C{i}"]["vocab_size"] + 1, (NB, B), generator=gen).to(DEV) for i in range(6)})
    big["label"] = (torch.rand(NB, B, generator=gen) < 0.3).float().to(DEV)
    batches = [{k: v[i].contiguous() for k, v in big.items()} for i in range(NB)]
    held = batches.pop()

    def train(replay, steps):
        torch.manual_seed(0)
        model = DeepFM(embedding_dim=16, hidden_units=[32, 16], enc_dict=enc).to(DEV)
        opt = FusedAdam(model.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay=replay)
        for i in range(steps):
            b = batches[i]
            model(b)["loss"].backward()
            opt.step()
            model.zero_grad()
        return model, opt

    def predict(model):
        model.eval()
        with torch.no_grad():
            return model(held, is_training=False)["pred"].cpu()

    # (1) one state, both replays
    model, opt = train("exact", 1000)
    lz = model.embedding_layer._lazy
    assert lz.t == 1000 and not lz.closed
    behind = 1000 - lz.last[lz.last > 0]
    assert int(behind.max()) > 700 and float(behind.float().mean()) > 100
    twin = copy.deepcopy(model)
    twin.embedding_layer._lazy.set_replay("closed")
    assert twin.embedding_layer._lazy.closed and twin.embedding_layer._lazy.last.data_ptr() != lz.last.data_ptr()
    p_ser, p_clo = predict(model), predict(twin)
    d1 = float((p_ser - p_clo).abs().max())
    rows_ser = model.embedding_layer.arena.detach().cpu()
    rows_clo = twin.embedding_layer.arena.detach().cpu()
    moved = (model.embedding_layer._lazy.last == 1000).cpu()
    assert int(moved.sum()) > 500
    d_rows = float((rows_ser[moved] - rows_clo[moved]).abs().max() / rows_ser.abs().max())
    print(f"\nsame state through both replays: max |pred diff| = {d1:.2e}, replayed rows differ by {d_rows:.2e} of the scale")
    assert d1 <= 1e-5 and d_rows <= 1e-6
    # (2) lockstep from the same start
    del model, opt, twin
    runs = []
    for replay in ("exact", "closed"):
        torch.manual_seed(0)
        mdl = DeepFM(embedding_dim=16, hidden_units=[32, 16], enc_dict=enc).to(DEV)
        runs.append((mdl, FusedAdam(mdl.parameters(), lr=1e-3, fuse_zero_grad=True, lazy_tables=True, replay=replay)))
    for i in range(1100):
        preds = []
        for mdl, o in runs:
            out = mdl(batches[i])
            out["loss"].backward()
            o.step()
            mdl.zero_grad()
            preds.append(out["pred"].detach())
        if i + 1 in (256, 300, 500, 700, 1100):
            d_pred = float((preds[0] - preds[1]).abs().max())
            d_dense = max(float((a.detach() - b.detach()).abs().max()) for (n, a), (_, b) in
                          zip(runs[0][0].named_parameters(), runs[1][0].named_parameters()) if "embedding" not in n)
            print(f"lockstep step {i + 1}: max |pred diff| {d_pred:.2e}, dense parameters {d_dense:.2e}")
            if i + 1 == 256:
                assert d_pred == 0.0 and d_dense == 0.0  # the closed form has not been used yet
            elif i + 1 <= 700:
                assert d_pred <= 1e-5 and d_dense <= 1e-6, (i + 1, d_pred, d_dense)
    assert runs[1][0].embedding_layer._lazy.closed and not runs[0][0].embedding_layer._lazy.closed
