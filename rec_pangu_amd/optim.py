"""FusedAdam — the optimiser RankTrainer.fit builds (reference: rec_pangu/trainer.py:75,
torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)), as fused HIP launches.

Semantics are the reference's: DENSE Adam over every parameter, embedding tables included — rows nobody
looked up still decay their moments and move.  Two ways to execute exactly that for the
arena-backed embedding tables (rec_pangu_amd.models.layers.EmbeddingLayer):

  lazy_tables=False  one flat dense launch over the p/g/m/v arenas per step (69 GB of HBM traffic per step at
                     Criteo shape — the optimiser then is 80 % of the train step);
  lazy_tables=True   exact lazy dense Adam (LazyAdamRows): a row's zero-gradient steps are replayed in
                     registers, with the dense kernel's own update function, when the row is next looked up or
                     flushed.  Bit-identical parameters and moments (tests/test_hip_lazy_adam.py), ~1/10 of
                     the traffic.  Anything that reads the raw tables (state_dict, .to, checkpoints) flushes
                     first; direct reads of `embedding.weight` need `optimizer.flush()`.

All other parameters go through one multi-tensor dense launch.  `fuse_zero_grad=True` clears gradients inside
the same pass (the model.zero_grad() that follows optimizer.step() in the reference loop,
model_pipeline.py:57-58).  CPU parameters are not handled here: make_adam builds torch.optim.Adam for those.
"""
import os
import weakref
from typing import Dict, List

import torch

from . import hip


class LazyAdamRows:
    """Per-EmbeddingLayer state of the lazy dense Adam: moment arenas, per-row `last` step stamps and the device table
    of per-step scalars {A_t, B_t} (computed by the C library, like the dense kernel does).

    replay="exact":  skipped zero-gradient steps are replayed one by one with the dense kernel's own update function:
                     bit-identical to dense execution (the parity mode);
    replay="closed": steps after CF_FROM are replayed in ONE evaluation per element whatever their number
                     (rp_lazy_adam_cf_table / adam.hip: uniformly convergent expansion of the summed updates, relative
                     truncation error <= 9e-8; the replay launch becomes an HBM stream instead of a VALU-bound serial
                     chain).  Within 1e-6 relative of the exact replay per replay (tests/test_hip_lazy_adam.py); only
                     for the reference's hyper-parameters betas = (0.9, 0.999), eps > 0 — anything else replays exactly."""
    TABLE_CHUNK = 1024
    CF_FROM = 256

    def __init__(self, store, betas, eps, owner=None, t0: int = 0, replay: str = "exact"):
        a = store.arena
        self.m, self.v = torch.zeros_like(a), torch.zeros_like(a)
        self.last = torch.zeros((a.shape[0],), dtype=torch.int32, device=a.device)
        self.betas, self.eps = betas, eps
        self.owner = owner           # weakref to the FusedAdam this state belongs to
        self.t = self.flushed_t = t0  # created mid-run (optimizer state loaded, arena re-packed): every row is current
        # device table of per-step scalars, indexed by ABSOLUTE step: row j = step j's {A_j, B_j};
        # row 0 is unused.  Rows <= t0 are never read (no row carries a stamp below t0) but must exist: the kernels
        # index the table with the step number.
        self._table = torch.zeros((t0 + 1, 2), dtype=torch.float32, device=a.device)
        self._table_lr = None
        self._table_from = t0 + 1
        # closed-form replay: {ns_j, d_j} = {-lr_j/(1-b1^j), 1/sqrt(1-b2^j)} by step in double (kept in either mode, so
        # that the mode can be switched mid-run), and the per-k coefficient table valid for replays that end at step
        # `_cf_for` (rebuilt on the device when the end step moves)
        self._ns_d = torch.zeros((t0 + 1, 2), dtype=torch.float64, device=a.device)
        self._cf_from = max(self.CF_FROM, t0)  # no row carries a stamp in (0, t0)
        self._cf, self._cf_for = None, -1
        self.set_replay(replay)

    def set_replay(self, replay: str):
        """switch between the serial ("exact") and the closed-form ("closed") replay; takes effect at the next replay"""
        assert replay in ("exact", "closed"), replay
        b = self.betas
        self.closed = (replay == "closed" and abs(b[0] - 0.9) < 1e-12 and abs(b[1] - 0.999) < 1e-12 and self.eps > 0)

    def apply(self, fn):
        self.m, self.v, self.last, self._table = fn(self.m), fn(self.v), fn(self.last), fn(self._table)
        self._ns_d = fn(self._ns_d)
        self._cf, self._cf_for = None, -1

    def _ensure_table(self, t_new, lr):
        """rows [.., t_new] of the scalar table must exist and row t_new must have been built with `lr`."""
        cap = self._table.shape[0] - 1
        if t_new <= cap and (self._table_lr == lr or t_new < self._table_from):
            return
        assert cap >= t_new - 1, f"lazy Adam step table has {cap + 1} rows, step {t_new} needs rows up to {t_new - 1}"
        hi = t_new + self.TABLE_CHUNK
        rows = [hip.adam_step_scalars(lr, self.betas[0], self.betas[1], s, self.eps) for s in range(t_new, hi + 1)]
        new = torch.tensor(rows, dtype=torch.float32, device=self._table.device)
        self._table = torch.cat([self._table[:t_new], new])  # steps < t_new keep the lr they were taken with
        j = torch.arange(t_new, hi + 1, dtype=torch.float64)
        ns = -float(lr) / (1.0 - float(self.betas[0]) ** j)
        d = 1.0 / torch.sqrt(1.0 - float(self.betas[1]) ** j)
        self._ns_d = torch.cat([self._ns_d[:t_new], torch.stack([ns, d], 1).to(self._ns_d.device)]).contiguous()
        self._table_lr, self._table_from = lr, t_new

    def _cf_args(self, t_end):
        """closed-form arguments of a replay that ends at step t_end: (table, cf_from), or (None, 0) = exact replay"""
        if not self.closed or t_end <= self._cf_from:
            return None, 0
        if self._cf_for != t_end:
            need = t_end - self._cf_from + 1
            if self._cf is None or self._cf.shape[0] < need:
                self._cf = torch.zeros((need + self.TABLE_CHUNK, 8), dtype=torch.float32, device=self.m.device)
            hip.lazy_adam_cf_table(self._ns_d, t_end, self._cf_from, self.betas[0], self.betas[1], self._cf)
            self._cf_for = t_end
        return self._cf, self._cf_from

    def _check_table(self, t_target):
        if self._table.shape[0] <= t_target:
            raise RuntimeError(f"lazy Adam: step table has {self._table.shape[0]} rows but step {t_target} is needed")

    def _sorted_touched(self, store):
        sk = store._touched
        if sk is None:
            return None
        if getattr(store, "_touched_unsorted", False):
            sk, _ = hip.sort_pairs(sk, end_bit=store._meta()[3])
        return sk

    def replay(self, store, sorted_keys):
        if self.t > 0:
            self._check_table(self.t)
            cf, cf_from = self._cf_args(self.t)
            hip.lazy_adam_rows(sorted_keys, store.embedding_dim, store.arena, None, self.m, self.v, self.last,
                               self._table, self.t, False, False, self.betas[0], self.betas[1], self.eps, cf, cf_from)

    def step(self, store, lr, zero_grad: bool = True):
        t_new = self.t + 1
        self._ensure_table(t_new, lr)
        self._check_table(t_new)
        sk = self._sorted_touched(store)
        if sk is not None and sk.numel():
            cf, cf_from = self._cf_args(self.t)  # the catch-up before the real step ends at t_new - 1
            hip.lazy_adam_rows(sk, store.embedding_dim, store.arena, store.grad_arena, self.m, self.v, self.last,
                               self._table, t_new, True, zero_grad, self.betas[0], self.betas[1], self.eps, cf, cf_from)
        self.t = t_new
        if zero_grad:  # FusedAdam(fuse_zero_grad=True): the gradient rows were cleared inside the step
            store.grads_were_zeroed()

    def flush(self, store):
        if self.flushed_t == self.t:
            return
        self._check_table(self.t)
        cf, cf_from = self._cf_args(self.t)
        hip.lazy_adam_flush(store.arena.shape[0], store.embedding_dim, store.arena, self.m, self.v, self.last,
                            self._table, self.t, self.betas[0], self.betas[1], self.eps, cf, cf_from)
        self.flushed_t = self.t


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, fuse_zero_grad=False,
                 lazy_tables=False, replay="exact"):
        if weight_decay != 0:
            raise ValueError("FusedAdam mirrors the reference's optimiser: weight_decay must be 0")
        if replay not in ("exact", "closed"):
            raise ValueError("replay must be 'exact' (bit-identical to dense execution) or 'closed' (closed-form replay)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        self.fuse_zero_grad = fuse_zero_grad
        self.lazy_tables = lazy_tables
        self.replay = replay
        self._arena_state: Dict[int, dict] = {}
        self._stores = {}
        self._plans: Dict[int, list] = {}

    @staticmethod
    def _store_of(p):
        ref = getattr(p, "_rp_store", None)
        return None if ref is None else ref()

    def set_replay(self, replay: str):
        """switch the lazy tables between the serial ("exact", bit-identical to dense execution) and the closed-form
        ("closed") replay mid-run, e.g. to finish a run in the parity mode"""
        if replay not in ("exact", "closed"):
            raise ValueError("replay must be 'exact' or 'closed'")
        self.replay = replay
        for store in self._stores.values():
            if store._lazy is not None:
                store._lazy.set_replay(replay)

    def flush(self):
        """Lazy mode: bring every embedding row to the current step (dense-equivalent state)."""
        for store in self._stores.values():
            store.flush_lazy()

    # The kernels keep the second moment as its square root (adam.hip: a zero-gradient step is then one multiply, which
    # is what the lazy replay is bound by).  Live state: 'exp_avg', 'exp_avg_sq_sqrt'.  state_dict() adds the squared
    # 'exp_avg_sq' torch.optim.Adam would hold (same keys, comparable / loadable there) and keeps the native tensor so
    # that a FusedAdam resume is bit-exact; load_state_dict() accepts either.
    SQRT_KEY = "exp_avg_sq_sqrt"

    def state_dict(self):
        self.flush()
        sd = super().state_dict()
        out_state = {}
        step_of = {}  # state index -> the step count of its group (torch.optim.Adam keeps one 'step' per parameter)
        for g in sd["param_groups"]:
            for i in g["params"]:
                step_of[i] = g.get("_rp_step", 0)
        for k, st in sd["state"].items():
            st = dict(st)
            if self.SQRT_KEY in st:
                st["exp_avg_sq"] = st[self.SQRT_KEY] * st[self.SQRT_KEY]
                st["step"] = torch.tensor(float(step_of.get(k, 0)))
            out_state[k] = st
        return {"state": out_state, "param_groups": sd["param_groups"]}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
            group["_rp_step"] = step = group.get("_rp_step", 0) + 1
            ps: List[torch.Tensor] = []
            gs: List[torch.Tensor] = []
            ms: List[torch.Tensor] = []
            vs: List[torch.Tensor] = []
            stores, arena_ok = {}, {}
            plan = self._plans.get(id(group))  # (param, its arena store or None), resolved once per parameter list
            if plan is None or len(plan) != len(group["params"]):
                plan = self._plans[id(group)] = [(p, self._store_of(p)) for p in group["params"]]
            for p, store in plan:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam only updates HIP-device parameters (use make_adam for CPU models)")
                if store is not None and getattr(p, "_rp_store", None) is None:
                    store = None  # the table was detached from its arena since the plan was made
                if store is not None:
                    sid = id(store)
                    if sid not in arena_ok:  # one check per layer, not per table
                        arena_ok[sid] = store.grads_are_arena()
                    if arena_ok[sid]:
                        stores[sid] = store
                        continue
                st = self.state[p]
                if not st:
                    st["exp_avg"], st[self.SQRT_KEY] = torch.zeros_like(p), torch.zeros_like(p)
                ps.append(p.data)
                gs.append(p.grad)
                ms.append(st["exp_avg"])
                vs.append(st[self.SQRT_KEY])
            for sid, store in stores.items():
                self._stores[sid] = store
                use_lazy = self.lazy_tables
                if use_lazy:
                    lz = store._lazy
                    mine = lz is not None and lz.owner is not None and lz.owner() is self
                    if lz is not None and not mine:
                        # the state another optimizer left on the layer (an earlier fit() on the same model): bring the
                        # rows to that optimizer's last step, then start over like a fresh torch.optim.Adam does
                        store.flush_lazy()
                        lz = None
                    if lz is None or lz.m.shape != store.arena.shape or lz.m.device != store.arena.device:
                        lz = store._lazy = LazyAdamRows(store, (b1, b2), eps, owner=weakref.ref(self), t0=step - 1,
                                                        replay=self.replay)
                        self._adopt_loaded_state(store, lz.m, lz.v, lz)
                        self._expose_state(store, lz.m, lz.v)
                    lz.step(store, lr, zero_grad=self.fuse_zero_grad)
                    continue
                st = self._arena_state.get(sid)
                if st is None or st["m"].shape != store.arena.shape or st["m"].device != store.arena.device:
                    st = {"m": torch.zeros_like(store.arena), "v": torch.zeros_like(store.arena)}
                    self._arena_state[sid] = st
                    self._adopt_loaded_state(store, st["m"], st["v"], None)
                    self._expose_state(store, st["m"], st["v"])
                ps.append(store.arena.view(-1))
                gs.append(store.grad_arena.view(-1))
                ms.append(st["m"].view(-1))
                vs.append(st["v"].view(-1))
                if self.fuse_zero_grad:
                    store.grads_were_zeroed()
            if ps:
                hip.adam_step(ps, gs, ms, vs, lr, b1, b2, eps, step, self.fuse_zero_grad)
        return loss

    def _adopt_loaded_state(self, store, m, v, lz):
        """Moments that load_state_dict() put into self.state for the table Parameters (optimizer resume) are copied
        into the freshly created moment arenas; with the lazy execution every row that has a moment is stamped
        current at the loaded step (rows without one stay at 0: a zero-gradient step is the identity for them)."""
        off, any_loaded = 0, False
        for p in store.table_parameters():
            r = p.shape[0]
            st = self.state.get(p, None)
            if st and "exp_avg" in st and st["exp_avg"].shape == p.shape and st["exp_avg"].data_ptr() != m[off:off + r].data_ptr():
                m[off:off + r].copy_(st["exp_avg"])
                v[off:off + r].copy_(st[self.SQRT_KEY])
                any_loaded = True
            off += r
        if any_loaded and lz is not None and lz.t > 0:
            live = (m != 0).any(dim=1) | (v != 0).any(dim=1)
            lz.last.copy_(live.to(torch.int32) * lz.t)

    def load_state_dict(self, state_dict):
        """torch semantics; the moments of arena-backed tables are adopted by the arena state at the next step()."""
        self.flush()
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            # a torch.optim.Adam state_dict carries the step count per parameter, not per group: without it the bias
            # correction (and the lazy step table) would restart at 1 on warm moments
            if "_rp_step" not in group:
                steps = [float(self.state[p]["step"]) for p in group["params"] if "step" in self.state.get(p, {})]
                group["_rp_step"] = int(max(steps)) if steps else 0
        for st in self.state.values():  # torch.optim.Adam layout -> native (sqrt of the second moment)
            if self.SQRT_KEY in st:
                st.pop("exp_avg_sq", None)
            elif "exp_avg_sq" in st:
                st[self.SQRT_KEY] = st.pop("exp_avg_sq").sqrt()
            st.pop("step", None)
        for store in list(self._stores.values()):
            store._lazy = None  # rebuilt from the loaded moments (and the loaded step count) at the next step()
        self._arena_state.clear()
        self._plans.clear()

    def _expose_state(self, store, m, v):
        """per-table views of the moment arenas, so optimizer.state / state_dict() look like torch.optim.Adam's"""
        off = 0
        for p in store.table_parameters():
            r = p.shape[0]
            self.state[p]["exp_avg"] = m[off:off + r]
            self.state[p].pop("exp_avg_sq", None)
            self.state[p][self.SQRT_KEY] = v[off:off + r]
            off += r


def make_adam(model, lr, lazy_tables=True, replay=None):
    """What RankTrainer.fit uses: fused HIP Adam for a HIP-resident model (lazy dense Adam on the embedding arenas by
    default), torch.optim.Adam on CPU (BASELINE config 0).  Hyper-parameters are the reference's (trainer.py:75).
    replay: how the lazy execution catches a row up — "closed" (default; closed-form replay, <= 1e-6 relative to the
    serial one per replay, see LazyAdamRows) or "exact" (serial replay, bit-identical to dense execution);
    the environment variable RP_LAZY_REPLAY overrides the default."""
    params = list(model.parameters())
    if params and params[0].is_cuda:
        if replay is None:
            replay = os.environ.get("RP_LAZY_REPLAY", "closed")
        return FusedAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, fuse_zero_grad=True,
                         lazy_tables=lazy_tables, replay=replay)
    return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
