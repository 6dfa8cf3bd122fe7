"""FusedAdam's state_dict layout vs torch.optim.Adam's (the optimizer the reference builds, rec_pangu/trainer.py:75): the
host-side conversion both ways — no kernel runs here (the stepping itself is covered on the device by
tests/test_hip_models.py::test_fused_adam_state_round_trips_with_torch_adam)."""
import torch

from rec_pangu_amd.optim import FusedAdam


def _params():
    g = torch.Generator().manual_seed(0)
    return [torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(3, generator=g))]


def test_torch_adam_state_loads_with_its_step_count():
    ps = _params()
    ref = torch.optim.Adam(ps, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for _ in range(3):
        for p in ps:
            p.grad = torch.ones_like(p)
        ref.step()
    fused = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-2)
    fused.load_state_dict(ref.state_dict())
    assert fused.param_groups[0]["_rp_step"] == 3, "the loaded step count must carry over (bias correction, lazy table)"
    for p, q in zip(fused.param_groups[0]["params"], ps):
        st = fused.state[p]
        assert set(st) == {"exp_avg", FusedAdam.SQRT_KEY}
        torch.testing.assert_close(st[FusedAdam.SQRT_KEY] ** 2, ref.state[q]["exp_avg_sq"])
        assert torch.equal(st["exp_avg"], ref.state[q]["exp_avg"])


def test_fused_state_dict_is_loadable_by_torch_adam():
    ps = _params()
    ref = torch.optim.Adam(ps, lr=1e-2)
    for p in ps:
        p.grad = torch.ones_like(p)
    ref.step()
    fused = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-2)
    fused.load_state_dict(ref.state_dict())
    sd = fused.state_dict()
    assert all(float(st["step"]) == 1.0 and "exp_avg_sq" in st for st in sd["state"].values())
    back = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-2)
    back.load_state_dict(sd)
    for p in back.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    back.step()  # raised KeyError('step') before
    assert float(back.state[back.param_groups[0]["params"][0]]["step"]) == 2.0


def test_step_tables_extend_in_place_and_follow_lr_changes(monkeypatch):
    """optim.StepTables (the per-step scalar tables a captured hipGraph holds the addresses of): rows are built a chunk
    ahead with the current lr, rebuilt IN PLACE from the current step on when lr changes (steps already taken keep theirs),
    and the buffers only move — `generation` counts it — when the capacity doubles.  Host-side logic: runs on CPU tensors
    (rp_adam_step_scalars is a host function of the library)."""
    import math
    from rec_pangu_amd import hip
    from rec_pangu_amd.optim import StepTables
    hip.lib()
    monkeypatch.setattr(StepTables, "MIN_CAPACITY", 0)  # (the production default leaves room for 64 k steps)
    b1, b2, eps = 0.9, 0.999, 1e-8
    tabs = StepTables((b1, b2), eps, torch.device("cpu"), t0=0, chunk=16)
    cap0, ptr0, gen0 = tabs.capacity, tabs.sc.data_ptr(), tabs.generation
    lr_at = {}
    lr = 1e-3
    for t in range(1, 120):
        if t in (7, 40, 41):
            lr *= 0.5
        assert not tabs.covers(t, lr) or tabs.lr == lr
        tabs.ensure(t, lr, chunk=16)
        assert tabs.covers(t, lr)
        lr_at[t] = lr
        if tabs.capacity == cap0:
            assert tabs.sc.data_ptr() == ptr0 and tabs.generation == gen0, "no reallocation before the capacity is exceeded"
    assert tabs.capacity > cap0 and tabs.generation > gen0
    for t in (1, 6, 7, 39, 40, 41, 42, 100, 119):
        a, b = hip.adam_step_scalars(lr_at[t], b1, b2, t, eps)
        assert float(tabs.sc[t, 0]) == a and float(tabs.sc[t, 1]) == b, f"row {t} must carry the lr step {t} was taken with"
        ns = -lr_at[t] / (1.0 - b1 ** t)
        d = 1.0 / math.sqrt(1.0 - b2 ** t)
        assert abs(float(tabs.ns_d[t, 0]) - ns) <= 1e-15 * abs(ns) + 1e-300 and abs(float(tabs.ns_d[t, 1]) - d) <= 1e-14 * d
    # rows ahead of the current step are provisional: they follow the next lr change
    tabs.ensure(120, 7e-4, chunk=16)
    a, _ = hip.adam_step_scalars(7e-4, b1, b2, 125, eps)
    assert float(tabs.sc[125, 0]) == a and float(tabs.sc[119, 0]) == hip.adam_step_scalars(lr_at[119], b1, b2, 119, eps)[0]
