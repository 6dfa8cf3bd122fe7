"""CPU restatement of how the build EXECUTES the reference's optimizer on the embedding tables — TEST INFRASTRUCTURE.

The reference trains with a dense `torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)` over every
parameter, the full `[V+1, D]` tables included (rec_pangu/trainer.py:75, stepped at rec_pangu/model_pipeline.py:57-58;
`nn.Embedding` is never sparse, rec_pangu/models/layers/embedding.py:31-34, so untouched rows get a zero gradient and
still decay their moments and move).  The HIP path keeps those semantics but does not stream 8.6 GB per step:

  lazy      a row's zero-gradient steps are executed when the row is next needed (csrc/adam.hip, "Exact LAZY dense Adam");
  deferred  (opt-in) so is its REAL step: the gradient row waits in the dense gradient arena, and one pass in front of the
            forward applies it, then the zero-gradient steps since (csrc/adam.hip, "DEFERRED execution").

`LazyAdamProtocol` below is that bookkeeping — the `last[row]` stamps, the pending encoding, catch-up / step / flush — in
plain float64 torch, with torch.optim.Adam's own update formulas applied per row and step.  In exact arithmetic the
protocol IS dense Adam; tests/test_oracle_optim.py checks it against `torch.optim.Adam` itself (the reference's optimizer
object) on random touch patterns with gradient accumulation, evaluation passes and mid-iteration flushes.  The GPU tests
(tests/test_hip_lazy_adam.py, tests/test_hip_deferred_adam.py) then hold the kernels to the dense kernel bit for bit.
Only tests import this module.
"""
from typing import Dict, Optional

import torch

Tensor = torch.Tensor


class LazyAdamProtocol:
    """One table arena `p [R, D]` under the lazy (and optionally deferred) execution of dense Adam.

    last[row] >= 0 : (p, m, v) of the row are current through step last[row]   (0 = never updated: m = v = 0, where a
                     zero-gradient step is the identity)
    last[row] <  0 : current through step l = -last[row] - 1, and grad[row] is the gradient of step l + 1, not applied yet
    """

    def __init__(self, p: Tensor, betas=(0.9, 0.999), eps: float = 1e-8, defer: bool = False):
        self.p = p.clone().double()
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.grad = torch.zeros_like(self.p)          # the dense gradient arena (aten::embedding_dense_backward)
        self.last = torch.zeros(p.shape[0], dtype=torch.int64)
        self.b1, self.b2 = betas
        self.eps = eps
        self.defer = defer
        self.t = 0                                    # completed optimizer steps
        self.lr_of: Dict[int, float] = {}             # the learning rate each step was taken with (optim.StepTables)
        self._marked_for = -1
        self._touched: Optional[Tensor] = None

    # -- torch.optim.Adam's single-tensor update for one row at step j (torch/optim/adam.py: _single_tensor_adam) ----
    def _real_step(self, row: int, j: int, g: Tensor) -> None:
        self.m[row] = self.m[row] + (g - self.m[row]) * (1 - self.b1)          # exp_avg.lerp_(grad, 1 - beta1)
        self.v[row] = self.v[row] * self.b2 + (1 - self.b2) * g * g            # exp_avg_sq.mul_().addcmul_()
        bc1, bc2 = 1 - self.b1 ** j, 1 - self.b2 ** j
        denom = self.v[row].sqrt() / bc2 ** 0.5 + self.eps
        self.p[row] = self.p[row] - (self.lr_of[j] / bc1) * self.m[row] / denom

    def _zero_steps(self, row: int, lo: int, hi: int) -> None:
        """steps lo+1 .. hi with a zero gradient"""
        zero = torch.zeros_like(self.p[row])
        for j in range(lo + 1, hi + 1):
            self._real_step(row, j, zero)

    def _owed(self, row: int, t_done: int, mark: bool) -> None:
        raw = int(self.last[row])
        pend = raw < 0
        l = -raw - 1 if pend else raw
        apply = pend and l + 1 <= t_done
        if apply:
            self._real_step(row, l + 1, self.grad[row].clone())
            self.grad[row] = 0
            l += 1
        behind = apply or (0 < l < t_done)
        if 0 < l < t_done:
            self._zero_steps(row, l, t_done)
        if pend and not apply:
            return                                    # the gradient of the step in progress keeps waiting
        if mark:
            self.last[row] = -(t_done + 1)
        elif behind:
            self.last[row] = t_done

    # -- the three moments of a training iteration -------------------------------------------------------------------
    def before_forward(self, rows: Tensor, grad_enabled: bool = True) -> Tensor:
        """bring the rows a forward is about to read up to date (LazyAdamRows.replay); returns their values"""
        for row in torch.unique(rows).tolist():
            if self.defer:
                self._owed(row, self.t, mark=grad_enabled)
            else:
                l = int(self.last[row])
                if 0 < l < self.t:
                    self._zero_steps(row, l, self.t)
                    self.last[row] = self.t
        if self.defer and grad_enabled:
            self._marked_for = self.t + 1
        return self.p[rows]

    def backward(self, rows: Tensor, g_rows: Tensor) -> None:
        """dense table gradient of one backward pass: sum over the lookups of each row (accumulates across passes)"""
        self.grad.index_add_(0, rows, g_rows.double())
        u = torch.unique(rows)
        self._touched = u if self._touched is None else torch.unique(torch.cat([self._touched, u]))

    def step(self, lr: float) -> None:
        """optimizer.step() followed by zero_grad() (FusedAdam(fuse_zero_grad=True))"""
        t_new = self.t + 1
        self.lr_of[t_new] = lr
        if self.defer and self._marked_for == t_new:
            pass                                      # the rows carry their stamps; the gradient rows wait where they are
        elif self._touched is not None:
            for row in self._touched.tolist():
                l = int(self.last[row])
                assert l >= 0 or -l - 1 == self.t, "an unstamped step on a row with an older waiting gradient"
                l = self.t if l < 0 else l
                if 0 < l < self.t:
                    self._zero_steps(row, l, self.t)
                self._real_step(row, t_new, self.grad[row].clone())
                self.grad[row] = 0
                self.last[row] = t_new
        self.t = t_new
        self._touched = None

    def flush(self) -> None:
        """every row through step t (state_dict(), checkpoints, .to())"""
        for row in range(self.p.shape[0]):
            raw = int(self.last[row])
            pend = raw < 0
            l = -raw - 1 if pend else raw
            apply = pend and l + 1 <= self.t
            if apply:
                self._real_step(row, l + 1, self.grad[row].clone())
                self.grad[row] = 0
                l += 1
            if 0 < l < self.t:
                self._zero_steps(row, l, self.t)
            if apply or (0 < l and not pend):
                self.last[row] = self.t
