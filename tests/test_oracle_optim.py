"""The lazy / deferred execution protocol of the table optimizer (oracle/ref_optim.py, a restatement of the bookkeeping
of csrc/adam.hip + rec_pangu_amd/optim.py) against the reference's optimizer object itself: a dense
torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) over the whole table (rec_pangu/trainer.py:75), float64.

In exact arithmetic both executions ARE dense Adam: every row, touched or not, takes every step.  What is checked here is
the protocol — stamps, waiting gradients, gradient accumulation, evaluation passes, flushes in the middle of an
iteration, a changing learning rate.  The kernels are held to the dense kernel bit for bit on the GPU
(tests/test_hip_lazy_adam.py, tests/test_hip_deferred_adam.py)."""
import pytest
import torch

from oracle.ref_optim import LazyAdamProtocol


def _schedule(R, steps, seed):
    """per step: rows of the training batch (with hot rows and repeats), their gradient rows, what else happens"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(steps):
        n = int(torch.randint(3, 12, (1,), generator=g))
        rows = torch.cat([torch.randint(0, 4, (3,), generator=g),             # hot rows: touched (nearly) every step
                          torch.randint(0, R, (n,), generator=g)])             # the long tail, many never touched
        grads = torch.randn(rows.numel(), 3, generator=g, dtype=torch.float64) * 10.0 ** float(-3 * torch.rand(1, generator=g))
        extra = None
        if i % 7 == 3:     # gradient accumulation: a second forward + backward before the step
            r2 = torch.randint(0, R, (5,), generator=g)
            extra = (r2, torch.randn(5, 3, generator=g, dtype=torch.float64))
        out.append({"rows": rows, "grads": grads, "extra": extra, "eval": torch.randint(0, R, (6,), generator=g) if i % 5 == 2 else None,
                    "flush": i in (11, 12, 30), "lr": 1e-2 * (1.0 + 0.5 * ((i // 9) % 3))})
    return out


@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("seed", [0, 1])
def test_protocol_is_dense_adam(defer, seed):
    R, D, steps = 300, 3, 45
    torch.manual_seed(seed)
    p0 = torch.randn(R, D, dtype=torch.float64)
    dense = torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([dense], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    lazy = LazyAdamProtocol(p0, defer=defer)
    for i, s in enumerate(_schedule(R, steps, seed + 10)):
        # -- the forward reads rows that must equal the dense state
        got = lazy.before_forward(s["rows"], grad_enabled=True)
        assert torch.allclose(got, dense.detach()[s["rows"]], rtol=1e-12, atol=1e-14), f"step {i}: forward rows"
        gd = torch.zeros(R, D, dtype=torch.float64).index_add_(0, s["rows"], s["grads"])
        lazy.backward(s["rows"], s["grads"])
        if s["extra"] is not None:
            r2, g2 = s["extra"]
            got2 = lazy.before_forward(r2, grad_enabled=True)
            assert torch.allclose(got2, dense.detach()[r2], rtol=1e-12, atol=1e-14), f"step {i}: second forward"
            lazy.backward(r2, g2)
            gd.index_add_(0, r2, g2)
        if s["eval"] is not None:   # an evaluation pass between backward and step (torch.no_grad())
            gote = lazy.before_forward(s["eval"], grad_enabled=False)
            assert torch.allclose(gote, dense.detach()[s["eval"]], rtol=1e-12, atol=1e-14), f"step {i}: eval rows"
        if s["flush"]:              # state_dict() while the gradients of the step in progress are waiting
            lazy.flush()
            assert torch.allclose(lazy.p, dense.detach(), rtol=1e-12, atol=1e-14), f"step {i}: flush before the step"
        for gr in ref.param_groups:
            gr["lr"] = s["lr"]
        dense.grad = gd
        ref.step()
        ref.zero_grad()
        lazy.step(s["lr"])
    if defer:
        assert int((lazy.last < 0).sum()) > 5, "no real step is waiting: the deferred branch did not run"
    never = int((lazy.last == 0).sum())
    assert never > 0, "the schedule must leave rows nobody ever looked up"
    lazy.flush()
    st = ref.state[dense]
    assert torch.allclose(lazy.p, dense.detach(), rtol=1e-11, atol=1e-13)
    assert torch.allclose(lazy.m, st["exp_avg"], rtol=1e-11, atol=1e-14)
    assert torch.allclose(lazy.v, st["exp_avg_sq"], rtol=1e-11, atol=1e-16)
    assert int((lazy.last < 0).sum()) == 0 and not bool(lazy.grad.any()), "a flush applies and clears every waiting gradient"
    assert torch.equal(lazy.p[lazy.last == 0], p0[lazy.last == 0]), "never-touched rows have not moved"
