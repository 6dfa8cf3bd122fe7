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
