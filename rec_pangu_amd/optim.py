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


class StepTables:
    """The per-step scalars of one Adam clock in PERSISTENT device buffers, indexed by absolute step (row 0 unused):
    sc [cap, 2] float32 = {A_t, B_t} (rp_adam_step_scalars), ns_d [cap, 2] float64 = {-lr_t/(1-b1^t), 1/sqrt(1-b2^t)}
    (closed-form replay), t_dev int32[1] = completed steps (read by the kernels when a step runs inside a captured
    hipGraph, graph_step.py).  Rows are built CHUNK steps ahead with the current lr and rebuilt in place from the
    current step on when lr changes; steps already taken keep the lr they were taken with.  The buffers only move when
    the capacity doubles (`generation` counts that: a captured graph holds their addresses)."""

    # Buffers start with room for this many steps (1.5 MB): a captured step holds their addresses, so every capacity
    # doubling costs a re-capture of both static input sets (a few ms) — with the round-3 start of ~2 k steps one of them
    # fell into the bench's timed window (+0.3 ms/step over 20 steps).  Tests lower it to exercise the doubling.
    MIN_CAPACITY = 65536

    def __init__(self, betas, eps, device, t0: int = 0, chunk: int = 1024):
        cap = max(t0 + 2 + 2 * chunk, self.MIN_CAPACITY)
        self.betas, self.eps = betas, eps
        self.sc = torch.zeros((cap, 2), dtype=torch.float32, device=device)
        self.ns_d = torch.zeros((cap, 2), dtype=torch.float64, device=device)
        self.t_dev = torch.full((1,), t0, dtype=torch.int32, device=device)
        self.filled_to, self.lr, self.lr_from = t0, None, t0 + 1
        self.generation = 0

    @property
    def capacity(self) -> int:
        return self.sc.shape[0]

    def apply(self, fn):
        self.sc, self.ns_d, self.t_dev = fn(self.sc), fn(self.ns_d), fn(self.t_dev)
        self.generation += 1

    def covers(self, t_new, lr) -> bool:
        return t_new <= self.filled_to and (self.lr == lr or t_new < self.lr_from)

    def ensure(self, t_new, lr, chunk: int = 1024, allow_gap: bool = False):
        """rows [.., t_new] exist and row t_new has been built with `lr`.  allow_gap: rows between the last one built and
        t_new were never needed and never will be (the DENSE table: a dense step reads its own row only, and eager dense
        steps read none — the table only moves when captured steps run); a lazy table is read for every skipped step
        and must be gap-free."""
        if self.covers(t_new, lr):
            return
        if allow_gap and self.filled_to < t_new - 1:
            self.filled_to = t_new - 1
            if t_new >= self.capacity:  # (the copy below keeps the old rows; the gap stays zero: never read)
                cap = max(2 * self.capacity, t_new + 1 + 2 * chunk)
                for name in ("sc", "ns_d"):
                    old = getattr(self, name)
                    new = torch.zeros((cap, 2), dtype=old.dtype, device=old.device)
                    new[:old.shape[0]].copy_(old)
                    setattr(self, name, new)
                self.generation += 1
        assert self.filled_to >= t_new - 1, f"step table filled to {self.filled_to}, step {t_new} needs {t_new - 1}"
        hi = t_new + chunk
        if hi >= self.capacity:
            cap = max(2 * self.capacity, hi + 1 + chunk)
            for name in ("sc", "ns_d"):
                old = getattr(self, name)
                new = torch.zeros((cap, 2), dtype=old.dtype, device=old.device)
                new[:old.shape[0]].copy_(old)
                setattr(self, name, new)
            self.generation += 1
        # (one C call for the whole chunk: the per-step loop over ctypes cost the host ~1 ms every 1024th step)
        self._upload(self.sc[t_new:hi + 1],
                     hip.adam_step_scalars_range(lr, self.betas[0], self.betas[1], t_new, hi + 1 - t_new, self.eps))
        j = torch.arange(t_new, hi + 1, dtype=torch.float64)
        ns = -float(lr) / (1.0 - float(self.betas[0]) ** j)
        d = 1.0 / torch.sqrt(1.0 - float(self.betas[1]) ** j)
        self._upload(self.ns_d[t_new:hi + 1], torch.stack([ns, d], 1))
        self.lr, self.lr_from, self.filled_to = lr, t_new, hi

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_pins":  # (staging buffers + their events: per instance, made on demand)
                setattr(new, k, copy.deepcopy(v, memo))
        return new

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k != "_pins"}

    def _upload(self, dst, src):
        """host rows -> device rows in stream order WITHOUT blocking the host: a copy from pageable memory waits for the
        stream to drain first — every 1024th step the host lost the 6 replayed steps it was ahead (5.4 ms inside one call,
        bench.py host_stall, round 5) and the device then idled until the next step was enqueued.  The pinned staging
        buffers are this table's own (round 6: a pinned allocation inside the call is a hipHostMalloc the first time — 5.5 ms
        where it fell on the first call of a timed window, profiles/r06_window_ramp.txt); a buffer is reused only once the
        copy that last read it has run (refills are a chunk of steps apart: the wait never blocks in practice)."""
        if dst.is_cuda:
            key = (src.dtype, tuple(src.shape))
            pins = self.__dict__.setdefault("_pins", {})
            slot = pins.get(key)
            if slot is None:
                if len(pins) > 8:
                    pins.clear()
                slot = pins[key] = [torch.empty(src.shape, dtype=src.dtype, pin_memory=True), None]
            pin, ev = slot
            if ev is not None:
                ev.synchronize()
            pin.copy_(src)
            dst.copy_(pin, non_blocking=True)
            ev = slot[1] = ev if ev is not None else torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dst.device))
        else:
            dst.copy_(src)


# RP_ADAM_NOCLEAR=0: the catch-up launch always clears the gradient rows it applies (the round-3 behaviour)
NOCLEAR = os.environ.get("RP_ADAM_NOCLEAR", "1") != "0"


class LazyAdamRows:
    """Per-EmbeddingLayer state of the lazy dense Adam: moment arenas, per-row `last` step stamps and the device tables
    of per-step scalars (StepTables).

    replay="exact":  skipped zero-gradient steps are replayed one by one with the dense kernel's own update function:
                     bit-identical to dense execution (the parity mode);
    replay="closed": steps after CF_FROM are replayed in ONE evaluation per element whatever their number
                     (rp_lazy_adam_cf_table / adam.hip: uniformly convergent expansion of the summed updates, relative
                     truncation error <= 9e-8; the replay launch becomes an HBM stream instead of a VALU-bound serial
                     chain).  Closer to float64 Adam than the serial fp32 replay (tests/test_hip_lazy_adam.py); only
                     for the reference's hyper-parameters betas = (0.9, 0.999), eps > 0 — anything else replays serially.

    device_clock (set by graph_step.GraphedTrainStep): the kernels read the step number from tabs.t_dev instead of the
    launch arguments, so that a captured hipGraph can be replayed; the host counter `t` is then advanced by the caller."""
    TABLE_CHUNK = 1024
    CF_FROM = 256

    def __init__(self, store, betas, eps, owner=None, t0: int = 0, replay: str = "exact", defer: bool = False):
        a = store.arena
        # defer: the real step of a row runs at its next touch, together with its zero-gradient steps (one launch per
        # training step instead of two; adam.hip "DEFERRED execution").  `last` then carries pending stamps (< 0).
        self.defer = defer
        self._marked_for = -1  # the step whose gradient rows the last catch-up launch stamped pending
        # (step, sorted keys) of a stamping launch that left the applied gradient rows UNCLEARED because the backward it
        # precedes overwrites them (rp_lazy_adam_catchup mark = 2): resolved by that backward (EmbeddingLayer.
        # accumulate_grad), or — when a second lookup or the optimizer step arrives first — by clearing those rows
        self._noclear = None
        self.m, self.v = torch.zeros_like(a), torch.zeros_like(a)
        self.last = torch.zeros((a.shape[0],), dtype=torch.int32, device=a.device)
        self.betas, self.eps = betas, eps
        self.owner = owner           # weakref to the FusedAdam this state belongs to
        self.t = self.flushed_t = t0  # created mid-run (optimizer state loaded, arena re-packed): every row is current
        # rows <= t0 of the tables are never read (no row carries a stamp below t0) but exist: the kernels index the
        # tables with the step number
        self.tabs = StepTables(betas, eps, a.device, t0, self.TABLE_CHUNK)
        self._cf_from = max(self.CF_FROM, t0)  # no row carries a stamp in (0, t0)
        # closed form: the coefficient table (adam.hip "TABLE LAYOUT": indexed by a row's stamp, final once the stamp is
        # ~372 steps old) is valid for replays that end at step `_cf_built`; -1 = a fresh buffer.  Bringing it to the next
        # step rebuilds only the youngest stamps: O(1) per training step whatever the step count
        self._cf, self._cf_built, self._cf_gen = None, -1, -1
        self._cf_in_capture = False  # a captured step holds this state's window rebuild (set by _cf_args under device_clock)
        self.device_clock = False
        self.set_replay(replay)

    _table = property(lambda self: self.tabs.sc)

    def set_replay(self, replay: str):
        """switch between the serial ("exact") and the closed-form ("closed") replay; takes effect at the next replay"""
        assert replay in ("exact", "closed"), replay
        b = self.betas
        self.closed = (replay == "closed" and abs(b[0] - 0.9) < 1e-12 and abs(b[1] - 0.999) < 1e-12 and self.eps > 0)

    def apply(self, fn):
        self.m, self.v, self.last = fn(self.m), fn(self.v), fn(self.last)
        self.tabs.apply(fn)
        self._cf, self._cf_built = None, -1

    def _ensure_table(self, t_new, lr):
        if self.device_clock:  # inside a captured step: graph_step prepared the tables before the launch
            assert self.tabs.covers(t_new, lr), "graphed step: the step tables were not prepared for this step / lr"
            return
        self.tabs.ensure(t_new, lr, self.TABLE_CHUNK)

    def _cf_buffer(self):
        if self._cf is None or self._cf_gen != self.tabs.generation or self._cf.shape[0] < self.tabs.capacity:
            self._cf = torch.zeros((self.tabs.capacity, 8), dtype=torch.float32, device=self.m.device)
            self._cf_gen, self._cf_built = self.tabs.generation, -1
        return self._cf

    def cf_sync(self, t_end):
        """host-clock build: the table is valid for replays that end at t_end (no launch when it already is)"""
        cf = self._cf_buffer()
        if self._cf_built > t_end:
            self._cf_built = -1  # the clock went BACK (an optimizer-state rollback): the youngest entries were built for a
            #                      later end step — a fresh build
        if self._cf_built != t_end and (t_end > self._cf_from or self._cf_built < 0):  # (a fresh buffer gets its power columns)
            hip.lazy_adam_cf_table(self.tabs.ns_d, t_end, self._cf_from, self.betas[0], self.betas[1], cf,
                                   built_to=self._cf_built)
            self._cf_built = t_end
        return cf

    def _cf_args(self, t_end, build: bool = True):
        """closed-form arguments of a replay that ends at step t_end: (table, cf_from), or (None, 0) = serial replay"""
        if not self.closed:
            return None, 0
        if self.device_clock:
            # inside a captured step: the launch reads the step on the device and rebuilds the window of stamps that are
            # not final yet; everything older was built before the capture (FusedAdam.prepare_step -> cf_sync) and by the
            # replays since (consecutive steps).  `_cf_built` is advanced by FusedAdam.advance_host().
            cf = self._cf_buffer()
            if build:
                hip.lazy_adam_cf_table(self.tabs.ns_d, self.tabs.capacity - 1, self._cf_from, self.betas[0], self.betas[1],
                                       cf, built_to=0, t_dev=self.tabs.t_dev)
                self._cf_in_capture = True  # this state's window rebuild is part of the captured step (advance_host)
            return cf, self._cf_from
        if t_end <= self._cf_from:
            return None, 0
        return self.cf_sync(t_end), self._cf_from

    def _check_table(self, t_target):
        if self.tabs.filled_to < t_target:
            raise RuntimeError(f"lazy Adam: step table filled to {self.tabs.filled_to} but step {t_target} is needed")

    def _sorted_touched(self, store):
        sk = store._touched
        if sk is None:
            return None
        if getattr(store, "_touched_unsorted", False):
            sk, _ = hip.sort_pairs(sk, end_bit=store._meta()[3])
        return sk

    def _t_dev(self):
        return self.tabs.t_dev if self.device_clock else None

    @staticmethod
    def _shadow_of(store):
        """the bf16 lookup copy of the store's tables (EmbeddingLayer.bf16_training), or None"""
        sh = getattr(store, "_shadow", None)
        return sh if (sh is not None and sh.shape == store.arena.shape and sh.device == store.arena.device) else None

    def replay(self, store, sorted_keys, mark=None):
        if self.defer:
            # everything the batch's rows are owed (pending real step + skipped steps); under autograd their gradient of
            # the step in progress is announced (they are stamped pending for step t + 1).  mark: given by callers that
            # run inside an autograd.Function.forward, where grad mode is off whatever the caller's was (sharded.py)
            if mark is None:
                mark = torch.is_grad_enabled()
            if self.t > 0 or mark:
                self._check_table(self.t)
                cf, cf_from = self._cf_args(self.t) if (self.t > 0 or self.device_clock) else (None, 0)
                mode = int(bool(mark))
                if mark:
                    # The clear of an applied gradient row is dead traffic (1 of the launch's 8 row transfers) when the
                    # backward that follows overwrites the row anyway: the store's gradient arena is "clean" (its next
                    # gradient launch does not accumulate) and this is the step's only stamping launch so far.  A second
                    # lookup before any backward breaks the promise: the first one's rows are cleared now.
                    clean = bool(getattr(store, "_noclear_ok", False)) and store.grad_arena is not None \
                        and bool(getattr(store, "_grad_clean", False)) and NOCLEAR
                    self.resolve_noclear(store)
                    if clean and self._marked_for != self.t + 1:
                        mode = 2
                hip.lazy_adam_catchup(sorted_keys, store.embedding_dim, store.arena, store.grad_arena, self.m, self.v,
                                      self.last, self.tabs.sc, self.t, mode, self.betas[0], self.betas[1], self.eps, cf,
                                      cf_from, self._t_dev(), shadow=self._shadow_of(store))
                if mark:
                    self._marked_for = self.t + 1
                    if mode == 2:
                        self._noclear = (self.t + 1, sorted_keys)
            return
        if self._shadow_of(store) is not None:
            raise RuntimeError("bf16-storage training (EmbeddingLayer.bf16_training) needs the deferred table optimizer: "
                               "make_adam(defer=True), the default")
        if self.t > 0:
            self._check_table(self.t)
            cf, cf_from = self._cf_args(self.t)
            hip.lazy_adam_rows(sorted_keys, store.embedding_dim, store.arena, None, self.m, self.v, self.last,
                               self.tabs.sc, self.t, False, False, self.betas[0], self.betas[1], self.eps, cf, cf_from,
                               self._t_dev())

    def resolve_noclear(self, store, written_keys=None):
        """a stamping launch left its applied gradient rows uncleared (mark = 2): `written_keys` is the sorted key list a
        non-accumulating gradient launch is about to overwrite — if it is that launch's own list the promise is kept;
        in every other case (another lookup, the optimizer step, another key list) the rows are cleared here"""
        pend, self._noclear = self._noclear, None
        if pend is None:
            return
        keys = pend[1]
        if written_keys is not None and (written_keys is keys or (written_keys.data_ptr() == keys.data_ptr()
                                                                    and written_keys.numel() == keys.numel())):
            return
        hip.zero_rows(keys, store.embedding_dim, store.grad_arena)

    def step(self, store, lr, zero_grad: bool = True):
        t_new = self.t + 1
        self._ensure_table(t_new, lr)
        self._check_table(t_new)
        self.resolve_noclear(store)  # (no backward consumed the promise: the stamped rows must read as zero gradients)
        sk = None if (self.defer and zero_grad and self._marked_for == t_new) else self._sorted_touched(store)
        if self.defer and zero_grad and self._marked_for == t_new:
            # the rows were stamped by the catch-up launch in front of their forward: their gradient rows stay where the
            # backward wrote them and are applied (with THIS step's scalars, sc[t_new]) at the rows' next touch
            pass
        elif sk is not None and sk.numel():
            # (deferred mode: the step right after this state was created — its forward ran before any stamp existed)
            cf, cf_from = self._cf_args(self.t, build=False)  # the catch-up before the real step ends at t_new - 1
            hip.lazy_adam_rows(sk, store.embedding_dim, store.arena, store.grad_arena, self.m, self.v, self.last,
                               self.tabs.sc, t_new, True, zero_grad, self.betas[0], self.betas[1], self.eps, cf, cf_from,
                               self._t_dev())
            if self._shadow_of(store) is not None:  # (an update that did not go through the deferred kernels)
                hip.rows_to_bf16(sk, store.embedding_dim, store.arena, store._shadow)
        if self.device_clock:
            owner = self.owner() if self.owner is not None else None
            if owner is not None and owner._in_step:
                owner._clock_ticks.append(self.tabs.t_dev)  # (one launch for every clock of the step: FusedAdam.step)
            else:
                hip.counter_add(self.tabs.t_dev, 1)
        self.t = t_new
        if zero_grad:  # FusedAdam(fuse_zero_grad=True): the gradient rows were cleared inside the step
            store.grads_were_zeroed()

    def flush(self, store):
        # (a stamping launch whose rows were never written by a backward — the catch-up ahead of a step that was not taken —
        #  left applied gradient rows uncleared: they must read as zero before anything applies them again)
        self.resolve_noclear(store)
        if self.flushed_t == self.t:
            return
        self._check_table(self.t)
        was, self.device_clock = self.device_clock, False  # (never inside a captured step: host arguments)
        cf, cf_from = self._cf_args(self.t)
        self.device_clock = was
        if self.defer:
            hip.lazy_adam_flush_deferred(store.arena.shape[0], store.embedding_dim, store.arena, store.grad_arena, self.m,
                                         self.v, self.last, self.tabs.sc, self.t, self.betas[0], self.betas[1], self.eps,
                                         cf, cf_from, shadow=self._shadow_of(store))
        else:
            hip.lazy_adam_flush(store.arena.shape[0], store.embedding_dim, store.arena, self.m, self.v, self.last,
                                self.tabs.sc, self.t, self.betas[0], self.betas[1], self.eps, cf, cf_from)
        self.flushed_t = self.t


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, fuse_zero_grad=False,
                 lazy_tables=False, replay="exact", defer=False):
        if weight_decay != 0:
            raise ValueError("FusedAdam mirrors the reference's optimiser: weight_decay must be 0")
        if replay not in ("exact", "closed"):
            raise ValueError("replay must be 'exact' (bit-identical to dense execution) or 'closed' (closed-form replay)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        self.fuse_zero_grad = fuse_zero_grad
        self.lazy_tables = lazy_tables
        self.replay = replay
        # defer (lazy tables only, needs the fused zero_grad: a separate zero_grad() would clear gradient rows that are
        # still waiting): LazyAdamRows(defer=True)
        self.defer = bool(defer and lazy_tables and fuse_zero_grad)
        self._device_clock = False  # graph_step.GraphedTrainStep: the kernels read the step number on the device
        self._dense_tabs: Dict[int, StepTables] = {}
        self._arena_state: Dict[int, dict] = {}
        self._stores = {}
        self._plans: Dict[int, list] = {}
        self._in_step = False      # inside step(): the device clocks of the step are advanced by ONE launch at its end
        self._clock_ticks: list = []

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

    def set_defer(self, on: bool):
        """switch the table optimizer between the immediate and the deferred execution of the real step mid-run, between
        two iterations (same results either way; LazyAdamRows).  Turning it off applies what is waiting: a flush."""
        on = bool(on and self.lazy_tables and self.fuse_zero_grad)
        if not on and self.defer:
            self.flush()
        self.defer = on
        for store in self._stores.values():
            lz = store._lazy
            if lz is not None:
                lz.resolve_noclear(store)
                lz.defer = on
                lz._marked_for = -1

    def zero_grad(self, set_to_none: bool = True):
        """torch semantics.  In-place zeroing (set_to_none=False) would wipe the gradient rows that still wait for their
        deferred step: they are applied first (a flush)."""
        if not set_to_none and self.defer:
            self.flush()
        return super().zero_grad(set_to_none=set_to_none)

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
        self._in_step, self._clock_ticks = True, []
        try:
            self._step_groups()
        except BaseException:
            # a step that raised part-way advanced the host counters of the groups it got through only; the device clocks
            # are ticked on SUCCESS only (ticking all of them in `finally` let device and host clocks diverge)
            self._clock_ticks = []
            raise
        finally:
            self._in_step = False
        ticks, self._clock_ticks = self._clock_ticks, []
        if self._side_dense():
            # catch-up ahead (graph_step): the table clocks tick on the main stream, in front of the next batch's catch-up; the
            # dense clocks behind the dense step, on the plan's inline side section
            table = {id(lz.tabs.t_dev) for lz in self._lazies()}
            mine = [t for t in ticks if id(t) in table]
            rest = [t for t in ticks if id(t) not in table]
            for i in range(0, len(mine), 8):
                hip.counters_add(mine[i:i + 8], 1)
            hip.LaunchPlan.section(2)
            try:
                for i in range(0, len(rest), 8):
                    hip.counters_add(rest[i:i + 8], 1)
            finally:
                hip.LaunchPlan.section(0)
            return loss
        for i in range(0, len(ticks), 8):  # (every lazy state and every parameter group has a clock of its own)
            hip.counters_add(ticks[i:i + 8], 1)
        return loss

    def _side_dense(self) -> bool:
        """the dense step of a recorded launch plan goes to the plan's inline side section (graph_step's catch-up ahead)"""
        return bool(getattr(self, "side_dense", False)) and self._device_clock and hip.LaunchPlan.is_recording()

    def _step_groups(self):
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
                                                        replay=self.replay, defer=self.defer)
                        self._adopt_loaded_state(store, lz.m, lz.v, lz)
                        self._expose_state(store, lz.m, lz.v)
                    lz.step(store, lr, zero_grad=self.fuse_zero_grad)
                    continue
                if getattr(store, "_shadow", None) is not None:
                    raise RuntimeError("bf16-storage training (EmbeddingLayer.bf16_training) needs FusedAdam(lazy_tables=True, "
                                       "defer=True): the dense table step does not maintain the bf16 lookup copy")
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
                tabs = self._dense_tabs.get(id(group)) if self._device_clock else None
                if tabs is not None:
                    assert tabs.covers(step, lr), "graphed step: the dense step table was not prepared for this step / lr"
                    side = self._side_dense()
                    if side:
                        hip.LaunchPlan.section(2)
                    try:
                        hip.adam_step(ps, gs, ms, vs, lr, b1, b2, eps, step, self.fuse_zero_grad, scalars=tabs.sc,
                                      t_dev=tabs.t_dev)
                    finally:
                        if side:
                            hip.LaunchPlan.section(0)
                    self._clock_ticks.append(tabs.t_dev)
                else:
                    hip.adam_step(ps, gs, ms, vs, lr, b1, b2, eps, step, self.fuse_zero_grad)

    # ---- device-resident step counters (graph_step.GraphedTrainStep) --------------------------------------------------
    def _lazies(self):
        return [st._lazy for st in self._stores.values() if st._lazy is not None]

    def set_device_clock(self, on: bool):
        """on: every kernel of step() (and the lazy replays of the forward) reads the step number from device memory
        (StepTables.t_dev) — what a captured hipGraph needs.  The host counters stay authoritative for everything outside
        the captured step; they must be advanced with advance_host() after every replay."""
        self._device_clock = on
        for group in self.param_groups:
            if on:
                t = group.get("_rp_step", 0)
                dev = next((p.device for p in group["params"] if p.is_cuda), None)
                tabs = self._dense_tabs.get(id(group))
                if tabs is None and dev is not None:
                    tabs = self._dense_tabs[id(group)] = StepTables(group["betas"], group["eps"], dev, t)
                if tabs is not None:
                    tabs.t_dev.fill_(t)
        for lz in self._lazies():
            lz.device_clock = on
            lz.tabs.t_dev.fill_(lz.t)

    def _lr_of(self, lz):
        """the learning rate of the parameter group that owns the tables of this lazy state"""
        for store in self._stores.values():
            if store._lazy is lz:
                tabs = store.table_parameters()
                for group in self.param_groups:
                    if tabs and any(p is tabs[0] for p in group["params"]):
                        return group["lr"]
        return self.param_groups[0]["lr"]

    def prepare_step(self):
        """tables of the NEXT step exist for the current lr (host work that cannot run inside a capture); returns a
        signature that changes when a captured step would no longer do what the eager step does: a buffer it holds has
        moved, or the launch sequence depends on a switch that was flipped since (closed / serial replay, deferred /
        immediate real step, fused zero_grad) — GraphedTrainStep re-captures on any change of it"""
        sig = [self.fuse_zero_grad, self.defer]
        for group in self.param_groups:
            tabs = self._dense_tabs.get(id(group))
            if tabs is None:
                dev = next((p.device for p in group["params"] if p.is_cuda), None)
                if dev is not None:
                    tabs = self._dense_tabs[id(group)] = StepTables(group["betas"], group["eps"], dev, group.get("_rp_step", 0))
            if tabs is not None:
                tabs.ensure(group.get("_rp_step", 0) + 1, group["lr"], allow_gap=True)
                sig.append(tabs.generation)
        for lz in self._lazies():  # (once per lazy state, with the lr of the group that owns its tables)
            lz.tabs.ensure(lz.t + 1, self._lr_of(lz), lz.TABLE_CHUNK)
            if lz.closed:
                lz._cf_buffer()
                # a captured step rebuilds the window of stamps that are not final yet (consecutive steps); anything the
                # window cannot reach — a fresh buffer, eager steps in serial mode since the last build — is built here
                if lz._cf_built < 0 or lz._cf_built < lz.t - 4:
                    lz.cf_sync(lz.t)
            sig.append((lz.tabs.generation, lz.closed, lz.defer, lz._cf_gen))
        return tuple(sig)

    def host_counters(self):
        return [g.get("_rp_step", 0) for g in self.param_groups], [lz.t for lz in self._lazies()]

    def set_host_counters(self, counters):
        for g, t in zip(self.param_groups, counters[0]):
            g["_rp_step"] = t
        for lz, t in zip(self._lazies(), counters[1]):
            lz.t = t

    def advance_host(self):
        """one captured step was replayed: the device counters moved, move the host ones"""
        for g in self.param_groups:
            g["_rp_step"] = g.get("_rp_step", 0) + 1
        for lz in self._lazies():
            if lz.closed and getattr(lz, "_cf_in_capture", False):
                # the replay in front of the captured forward brought the table to this step — only where that rebuild was
                # recorded in the capture: a state whose layer the captured step never looks up keeps its old mark and is
                # rebuilt by the next eager cf_sync
                lz._cf_built = lz.t
            lz.t += 1
            if lz.closed and getattr(lz, "_cf_in_capture", False) and getattr(lz, "_cf_ahead", False):
                lz._cf_built = lz.t  # (catch-up ahead: the rebuild recorded BEHIND the step brought the table to the new step)

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
        if lz is not None:
            lz._cf_built = -1  # (whatever the closed-form table was built for, the loaded clock may differ)

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


def make_adam(model, lr, lazy_tables=True, replay=None, defer=None):
    """What RankTrainer.fit uses: fused HIP Adam for a HIP-resident model (lazy dense Adam on the embedding arenas by
    default), torch.optim.Adam on CPU (BASELINE config 0).  Hyper-parameters are the reference's (trainer.py:75).
    replay: how the lazy execution catches a row up — "closed" (default; closed-form replay, <= 1e-6 relative to the
    serial one per replay, see LazyAdamRows) or "exact" (serial replay, bit-identical to dense execution);
    the environment variable RP_LAZY_REPLAY overrides the default.
    defer: run a row's real step at its next touch, in the one launch that also replays its skipped steps (identical
    results after a flush — tests/test_hip_deferred_adam.py; one optimizer launch per step on the tables instead of two).
    Table .grad rows then hold gradients that are still waiting after step(), so anything that READS or rewrites table
    gradients between backward and step (gradient clipping on the tables) must use defer=False — and is told so: a table's
    .grad is a DeferredGradView in this mode, every torch operation on it raises (models/layers/embedding.py;
    tests/test_hip_deferred_adam.py::test_reading_table_gradients_under_the_deferred_step_raises).  Default ON since round 4
    (the whole -m gpu suite runs in it; environment variable RP_ADAM_DEFER=0 turns it off)."""
    params = list(model.parameters())
    if params and params[0].is_cuda:
        if replay is None:
            replay = os.environ.get("RP_LAZY_REPLAY", "closed")
        if defer is None:
            defer = os.environ.get("RP_ADAM_DEFER", "1") != "0"
        return FusedAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, fuse_zero_grad=True,
                         lazy_tables=lazy_tables, replay=replay, defer=defer)
    return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
