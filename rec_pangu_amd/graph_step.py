"""A whole training step — forward, backward, optimizer step, zero_grad — captured once and replayed: as a LAUNCH PLAN
(the library's own recorder, csrc/plan.hip: the recorded launches are re-issued with hipLaunchKernel — the device sees the
eager stream of kernels, the host spends a few microseconds per launch, and the next batch's row sort runs on a side stream
beside the step) when the captured step holds nothing but library launches, else as a hipGraph (backend="hipgraph", or a
step with ATen kernels / memsets in it: DCN's weight-space cumsum, a model with active dropout ...).

Why: the eager step of the headline configuration is 13 kernel launches + the 9 launches of the row sort behind autograd,
ctypes and torch.empty: ~1.0 ms of host time per step against ~1.3 ms on the device — and at the per-GPU batch of a
strong-scaling run (8192) the host IS the step time (1.10 ms eager against 0.32 ms replayed, profiles/microbench/probes/probe_graph.py).
Replaying a captured graph costs the host one launch (~0.1 ms) and the device no inter-kernel gaps of host origin.

What makes a step capturable here:
  * the STEP NUMBER lives on the device: every launch argument is frozen at capture, so the Adam kernels read the step from
    StepTables.t_dev (rp_adam_step / rp_lazy_adam_rows / rp_lazy_adam_cf_table with t_dev, rp_counter_add) and the
    per-step scalar tables are persistent buffers that the host extends in place (optim.StepTables);
  * STATIC INPUT BUFFERS, two sets: graph P reads the current batch from X[P] and its (keys, sorted keys, positions) from
    persistent tensors (EmbeddingLayer.pin_sort); the NEXT batch is copied into X[1-P] (one multi-tensor copy per dtype)
    before the launch and sorted at the END of graph P, on the same stream, into the tensors graph 1-P reads.  The sort
    is NOT overlapped with the step as the eager path does it (side stream): a fork inside one graph is replayed without
    overlap by this runtime, and a second graph on a second stream (measured 0.49 against 0.52 ms at b = 8192) died with
    memory access faults whenever the host ran more than a few steps ahead without a device-wide synchronisation
    (profiles/microbench/probes/probe_graph6.py) — one linear graph on one stream has run thousands of unsynchronised steps;
  * no host synchronisation inside the step: check_indices = "deferred" (raise_if_bad_index() after the run).

The captured launches are the eager path's own (same kernels, same order, same arguments except the step number's
source): results are bit-identical to the eager loop (tests/test_hip_graph.py).  Uses torch.cuda.CUDAGraph (= hipGraph
on ROCm) for the capture and for the allocator's graph-private pool; the first `eager_steps` calls run eagerly so that
optimizer state, gradient buffers and caches exist before the capture.
"""
import os
from typing import Callable, Dict, Optional

import torch

from . import hip
from .models.layers import embedding as _emb

# the run-ahead bound's completion markers: hip.Marker (no system-scope fence at the record) unless RP_STEP_MARKERS=torch
_TORCH_MARKERS = os.environ.get("RP_STEP_MARKERS", "") == "torch"



class GraphedTrainStep:
    """step = GraphedTrainStep(model, optimizer); out = step(batch, next_batch) in the training loop.

    `next_batch` is the batch of the following call (None at the end of an epoch: that step then runs eagerly).  The
    returned dict holds detached, STATIC tensors ('pred', 'loss', ...) that the next call on the same graph overwrites
    (do not keep an autograd graph of an earlier eager step alive across the first captured call: its AccumulateGrad nodes
    are bound to the eager stream).  `post_backward`: a hook run
    between backward and optimizer step (eager and captured alike).

    CONTRACT for the batches (launch plans re-point the recorded launches at the caller's own tensors, no staging copy):
    a batch handed in as `next_batch` is read TWICE — by that call's replay (its keys + row sort, a step ahead) and by the next
    call's (the step itself) — and both reads happen on the device after the calls have returned.  Its tensors must
    therefore stay alive and UNCHANGED from the call that announces them until MAX_IN_FLIGHT + 2 further calls have been
    made (the step holds references that long).  A loader that refills a device buffer in place is detected where torch can
    see it — the tensor's data pointer or version counter differs from what was announced — and the batch is then staged
    and re-sorted on the spot (correct, one copy + one sort slower); a refill torch cannot see (a raw kernel writing through
    data_ptr()) is the caller's responsibility.  RP_PLAN_REBIND=0 restores the staging copy (a snapshot at call time)."""

    MAX_IN_FLIGHT = 6

    def __del__(self):
        try:
            if any(g is not None for g in self.graphs):
                torch.cuda.synchronize()
            for pl in self.plans:
                if pl is not None:
                    pl.destroy()
        except Exception:
            pass

    def __init__(self, model, optimizer, post_backward: Optional[Callable[[], None]] = None, eager_steps: int = 2,
                 backend: Optional[str] = None):
        if not getattr(model, "on_hip", False):
            raise RuntimeError("GraphedTrainStep needs a HIP-resident model")
        if not hasattr(optimizer, "set_device_clock"):
            raise RuntimeError("GraphedTrainStep needs rec_pangu_amd.optim.FusedAdam (device-resident step counters)")
        for m in model.modules():
            if getattr(m, "check_indices", "deferred") != "deferred":
                raise RuntimeError("GraphedTrainStep: set check_indices = 'deferred' on the embedding layers (a captured "
                                   "step cannot synchronise with the host; call raise_if_bad_index() after the run)")
        # Row-sharded tables (sharded.ShardedEmbeddingLayer, round 4): the step is captured WITH its collectives — in the
        # fixed-capacity exchange every split size is a constant, so RCCL's all-to-alls and the dense all-reduce
        # (post_backward = lambda: allreduce_dense_grads(model)) become graph nodes.  Such a step always replays as a
        # hipGraph (the collectives are not the library's launches); the route of a batch is built inside its own step
        # (no look-ahead: one graph, one stream).  Every rank must capture and replay in lockstep.
        self._sharded = not hasattr(model.embedding_layer, "pin_sort")
        if self._sharded and not hasattr(model.embedding_layer, "local_arena"):
            raise RuntimeError("GraphedTrainStep: unknown embedding layer type")
        self.model, self.opt, self.post_backward = model, optimizer, post_backward
        # "plan": replay the captured step as a launch plan when it holds only library launches (checked against the
        # captured hipGraph's nodes), else as the hipGraph; "hipgraph": always the hipGraph
        self.backend = backend or os.environ.get("RP_GRAPH_BACKEND", "plan")
        if self.backend not in ("plan", "hipgraph"):
            raise ValueError("backend must be 'plan' or 'hipgraph'")
        self.plans = [None, None]
        self.backend_used = None     # "plan" | "hipgraph" once the first step has been captured
        self.why_not_plan = None
        self.eager_left = eager_steps
        self.X = None            # the two static batches
        self.graphs = [None, None]
        self.outs = [None, None]
        self.P = 0
        self._inflight, self._ev_pool = [], []  # completion events of the replays not yet known to have finished
        self._staged = None      # the caller's batch object whose content X[P] holds (sorted and pinned)
        self._staged_sig = None  # (data_ptr, version) of its tensors when it was announced (a recycled buffer re-stages)
        self._sig = None
        self._pool = None
        self._dev = None         # host mirror of the device counters
        self._one = None         # the seed gradient of captured backward passes (see _seed_grad)
        self._seed, self._seed_value = None, 1.0
        # active dropout inside the step (the reference's default for xDeepFM / AutoInt / MMOE): the generator offset of a
        # step's start lives on the device (`_drop_clock`, advanced by the step's last launch), every dropout launch adds
        # its place in the step; the host generator is advanced after each replay, so eager and captured steps draw from
        # ONE stream of offsets (hip._dropout_seed_offset, csrc/dropout.hip: rp_dropout_fwd_dev)
        self._drop_clock = None
        self._drop_clock_host = None   # the value the device clock holds (will hold once the queued replays have run)
        self._drop_calls = [0, 0]      # dropout launches in the captured step of each static batch
        self._drop_seed = None
        self.replays = 0
        self.captures = 0
        # host time of the replay calls: `host_call_s` = inside __call__ WITHOUT the back-pressure wait (`host_wait_s`: the
        # host blocks there once MAX_IN_FLIGHT replays are queued — device time, not host work)
        self.host_call_s = self.host_wait_s = 0.0
        self.host_seg_max = {}         # slowest host time per part of a replay call (copy / prepare / replay / record / advance)
        self.host_call_max_s = 0.0     # the slowest single call (reset by the caller: bench.py's timed window)
        self._last_plan = None
        self._copy_lists = None
        # Catch-up ahead (launch plans, deferred lazy table optimizer): the captured step ends with the optimizer catch-up of
        # the NEXT batch's rows — the step's forward then starts at the lookup, the dense Adam launch and the rest of the
        # first layer's side work run BESIDE that catch-up instead of in front of it.  `_ahead_valid`: the static batch whose
        # rows the last call left caught up and stamped (None: nobody's); `_ahead_noclear[P]`: plan P's catch-up left its
        # applied gradient rows uncleared for the next backward to overwrite (LazyAdamRows.replay, mark = 2).
        # Input rebinding (launch plans): instead of copying every batch into a static input buffer, the recorded launch
        # arguments that point at those buffers are re-pointed at the batch's own tensors before each replay
        # (rp_plan_bind_inputs / _set_inputs).  `_x_valid[Q]`: static batch Q holds the content of the batch it stands for
        # (false after a step that read the tensors directly); `_held`: the batches of the replays that may still be in flight.
        self.rebind = os.environ.get("RP_PLAN_REBIND", "1") != "0"
        self.bind_report = None   # {"sites": {input: argument words}, "interior": derived pointers} of the last capture
        self._bind_sites = [0, 0]
        self._x_valid = [False, False]
        self._held = []
        self.ahead = os.environ.get("RP_CATCHUP_AHEAD", "0") == "1"
        self._ahead_used = False
        self._ahead_valid = None
        self._ahead_noclear = [False, False]

    # ---- catch-up ahead --------------------------------------------------------------------------------------------------
    def _ahead_lazy(self):
        """the embedding layer's deferred lazy state when the step can catch up ahead, else None"""
        if not self.ahead or self._sharded or self.backend != "plan":
            return None
        lz = getattr(self.model.embedding_layer, "_lazy", None)
        return lz if (lz is not None and lz.defer and lz.t > 0 and getattr(self.opt, "defer", False)) else None

    def _ahead_drop(self):
        """the promise of the last replay is not going to be kept by a replay (an eager step follows, the static batch is
        restaged, the captures are dropped): rows it stamped whose applied gradient rows were left uncleared read as zero
        gradients from here on — what LazyAdamRows does when a second lookup precedes the backward"""
        if self._ahead_valid is not None:
            emb = self.model.embedding_layer
            lz = getattr(emb, "_lazy", None)
            if lz is not None:
                lz.resolve_noclear(emb)
            emb._ahead_done = None
            self._ahead_valid = None

    # ---- pieces --------------------------------------------------------------------------------------------------------
    def _eager(self, batch, nxt):
        self._ahead_drop()
        if nxt is not None:
            self.model.prefetch(nxt)
        out = self.model(batch)
        out["loss"].backward()
        if self.post_backward is not None:
            self.post_backward()
        self.opt.step()
        self.model.zero_grad()
        # detached: an autograd graph kept alive by a returned loss would keep its AccumulateGrad nodes (bound to THIS
        # stream) alive into the capture, which runs on another stream — that cross-stream dependency breaks the capture
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}

    # Only with RP_SORT=rocprim: rocPRIM switches to its onesweep radix sort above ~1 M pairs (global digit counters reset
    # by memset nodes between the passes).  Replayed from a graph WITHOUT device-wide synchronisations in between, that path
    # ends in memory access faults within a few hundred steps (B = 40960 x 26 fields = 1.06 M pairs: fault; B = 32768 =
    # 0.85 M pairs: 600 steps clean; B = 65536 with torch.cuda.synchronize() every 50 steps: 1300 steps clean).  The own
    # radix sort of csrc/sort.hip (the default; kernels only, no memset nodes) replays cleanly at any size — B = 40960 and
    # B = 65536, 1500 unsynchronised replays each, profiles/microbench/probes/probe_graph6.py — which is what pinned the faults on that path.
    # (Captured steps remain the tool for SMALL, host-bound batches: at B = 65536 a replay is slower than eager launches.)
    MAX_PAIRS_ROCPRIM = 900_000

    def _alloc(self, batch):
        n_pairs = sum(batch[c].numel() for c in self.model.embedding_layer.emb_feature)
        if os.environ.get("RP_SORT") == "rocprim" and n_pairs > self.MAX_PAIRS_ROCPRIM:
            raise RuntimeError(f"GraphedTrainStep: {n_pairs} (sample, field) pairs per batch — above {self.MAX_PAIRS_ROCPRIM} "
                               "rocPRIM's row sort takes its onesweep path, which does not survive unsynchronised graph "
                               "replays on this runtime; unset RP_SORT=rocprim or run batches of this size eagerly")
        self.X = [{k: torch.zeros_like(v) for k, v in batch.items()} for _ in range(2)]
        self._one = torch.ones((), dtype=torch.float32, device=next(iter(batch.values())).device)
        self._drop_clock = torch.zeros((1,), dtype=torch.int64, device=self._one.device)
        self._keys = list(batch.keys())
        self._copy_lists = [None, None]  # per static batch: the destination side of its staging copy, converted once
        self._x_valid, self._bind_sites, self._held = [False, False], [0, 0], []
        self._pins = None  # (sharded: the static batches' prepared lookups in persistent buffers, made once the capacity is known)
        if self._sharded:
            return
        for x in self.X:  # both static batches get their persistent sort buffers before anything is captured
            self.model.embedding_layer.pin_sort(x)

    def _copy(self, P, batch):
        cl = self._copy_lists[P] if self._copy_lists is not None else None
        if cl is None and self._copy_lists is not None:
            cl = self._copy_lists[P] = hip.CopyList([self.X[P][k] for k in self._keys])
        src = [batch[k] for k in self._keys]
        if cl is not None and cl(src):
            return
        dst = [self.X[P][k] for k in self._keys]
        # ONE launch for the ~40 columns of a batch: torch._foreach_copy_ issues a device-to-device copy per tensor here, and
        # 40 small copies cost the stream ~0.25 ms per step in dispatch gaps (rocprofv3: 42 x __amd_rocclr_copyBuffer per
        # step — what round 3 had read as "~10 us per graph node")
        if not hip.multi_copy(dst, src):
            torch._foreach_copy_(dst, src)

    def _sig_of(self, batch):
        """what torch can see of a batch's buffers: (address, version counter) per tensor"""
        return [(t.data_ptr(), t._version) for t in batch.values()]

    def _stage_current(self, batch):
        self._ahead_drop()  # (before the pinned key list it names is re-sorted in place)
        self._copy(self.P, batch)
        self._x_valid[self.P] = True
        if not self._sharded:
            self.model.embedding_layer.pin_sort(self.X[self.P])
        elif self._sharded_pins():
            # the route, id exchange and owner-side sort of the batch, eagerly (no replay prepared them beside the step before)
            self.model.embedding_layer.route_into(self.X[self.P], self._pins[self.P])
        self._staged = batch

    def _sharded_pins(self) -> bool:
        """round 6: a row-sharded step reads the part of its lookup that does not depend on the weights (route, requested rows,
        owner-side sort) from persistent buffers, filled by the replay of the step BEFORE it on the ahead stream
        (ShardedEmbeddingLayer.route_into inside the recording) — RP_SHARD_AHEAD=0: built inside the step itself (rounds 4-5)"""
        if os.environ.get("RP_SHARD_AHEAD", "1") == "0" or self.backend != "plan":
            return False
        emb = self.model.embedding_layer
        if self._pins is not None and self._pins[0].capacity != emb._capacity:
            torch.cuda.synchronize()  # (the exchange capacity was re-measured: the recorded steps hold the old buffers)
            self._drop_captures()
            self._pins = None
        if self._pins is None:
            pins = [emb.pinned_route(x) for x in self.X]
            self._pins = pins if all(p is not None for p in pins) else None
        return self._pins is not None

    def _capture(self, P, force_graph: bool = False):
        if self.graphs[1 - P] is None:
            # the flags describe the captures in use: with none left, a layer this capture no longer looks up must not keep
            # "a captured step rebuilds my window" from an earlier one (the capture below sets them again)
            for lz in self._lazy_states():
                lz._cf_in_capture = False
        counters = self.opt.host_counters()
        self.opt.set_device_clock(True)
        self._dev = counters
        torch.cuda.synchronize()
        # Row-sharded steps (round 6): recorded as a launch plan too — the collectives are NOT issued under the capture, the
        # plan is cut at each of them (sharded._recorded -> rp_plan_host_mark) and a replay issues them itself between two
        # segments.  A step that turns out to hold foreign launches is captured AGAIN, collectives included, as a hipGraph.
        want_plan = self.backend == "plan" and not force_graph
        world = getattr(self.model.embedding_layer, "world", 1) if self._sharded else 1
        seed_scaled = want_plan and self._sharded and world > 1
        if seed_scaled and (self._seed is None or self._seed_value != 1.0 / world):
            self._seed, self._seed_value = torch.full_like(self._one, 1.0 / world), 1.0 / world  # (outside the capture)
        # (keep_graph: the hipGraph_t stays inspectable — rp_graph_node_counts — and is only instantiated if it is replayed)
        g = torch.cuda.CUDAGraph(keep_graph=True) if want_plan else torch.cuda.CUDAGraph()
        plan = hip.LaunchPlan() if want_plan else None
        gen = hip.device_generator(self._drop_clock.device)
        drop = hip.DROPOUT_CAPTURE[0] = {"seed": gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, "clock": self._drop_clock, "calls": 0}
        emb = self.model.embedding_layer
        ahead = plan is not None and self._ahead_lazy() is not None and self._ahead_valid == P
        hip.LaunchPlan.ahead = ahead
        self.opt.side_dense = ahead
        held = []
        try:
            if plan is not None:
                plan.begin()
            try:
                # (sharded: the process group's watchdog thread polls the events of earlier, eager collectives while this
                # thread captures — legal only in thread-local capture mode; in the default global mode its hipEventQuery
                # aborts the process with hipErrorStreamCaptureUnsupported)
                mode = "thread_local" if self._sharded else "global"
                from .sharded import seed_scaled as _seed_scaled
                with torch.cuda.graph(g, pool=self._pool, capture_error_mode=mode), _seed_scaled(seed_scaled):
                    fork_start = os.environ.get("RP_PLAN_FORK", "start" if ahead else "backward") == "start"
                    if plan is not None and fork_start:
                        # (A/B switch: the next batch's sort beside the catch-up and the gather instead.  Catch-up ahead: the
                        #  step itself consumes the sort at its end and starts at the lookup — the sort gets the whole step)
                        plan.fork_here()
                    pins = self._pins if (self._sharded and self._pins is not None) else None
                    if pins is not None:
                        # the NEXT batch's route, id exchange and owner-side sort into its persistent buffers: segments of the
                        # ahead stream (LaunchPlan.cut), forked at the start of a replay and joined at its end.  Recorded FIRST,
                        # every temporary held until the recording ends: they run BESIDE the step, whose temporaries must not
                        # land in memory the capture's allocator got back from them (hip.holding)
                        hip.LaunchPlan.cut(1)
                        with hip.holding() as held_ahead:
                            emb.route_into(self.X[1 - P], pins[1 - P])
                        held.append(held_ahead)
                        hip.LaunchPlan.cut(0)
                        emb.use_pinned(pins[P])
                    out = self.model(self.X[P])  # (finds X[P]'s pinned sort; nothing is announced inside the capture)
                    if plan is not None and not fork_start:
                        # the side section (the next batch's sort, recorded last) is forked here: beside the backward,
                        # where the eager path starts it too — beside the catch-up and the gather it costs more than it hides
                        plan.fork_here()
                    # the seed gradient is a persistent 1.0: loss.backward() would create it with an ATen fill kernel — the
                    # one launch of a DeepFM step that is not the library's (a launch plan must hold them all)
                    out["loss"].backward(gradient=self._seed_grad(out["loss"], 1.0 / world if seed_scaled else 1.0))
                    if plan is not None:
                        hip.LaunchPlan.settle()
                    if self.post_backward is not None:
                        self.post_backward()
                    self.opt.step()
                    if ahead:
                        # the dense step runs on the side section until the end of the replay: the capture's one-stream
                        # allocator must not hand the gradients' memory to the launches recorded behind it
                        held += [p.grad for g_ in self.opt.param_groups for p in g_["params"] if p.grad is not None]
                    self.model.zero_grad()
                    # the next batch (already staged in X[1-P] when the step is launched): keys + sort into its pinned
                    # tensors.  It depends on nothing the step computes and touches persistent buffers only: a plan
                    # re-issues these launches on its side stream, beside the step (section 1)
                    if drop["calls"] > 0:  # the step's last launch on the main stream: the dropout clock of the next step
                        hip.counter_add_u64(self._drop_clock, 4 * drop["calls"])
                    if plan is not None:
                        plan.section(1)
                    if not self._sharded:
                        self.model.embedding_layer._sort_into(self.X[1 - P], self._pinned(1 - P), on_side_stream=False)
                    if plan is not None:
                        plan.section(0)
                    if ahead:
                        plan.join_side()  # the main stream reads the next batch's sorted keys from here on
                        if not emb.catch_up_ahead(self._pinned(1 - P)[1]):
                            raise RuntimeError("GraphedTrainStep: the catch-up ahead was not issued")
            finally:
                hip.DROPOUT_CAPTURE[0] = None
                hip.LaunchPlan.ahead = False
                self.opt.side_dense = False
                if plan is not None:
                    plan.end()
                del held
        finally:
            # the capture ran the python of one step without executing a kernel: put the host counters back
            self.opt.set_host_counters(counters)
            self.opt.set_device_clock(False)
        had_pool = self._pool is not None
        if self._pool is None:
            self._pool = g.pool()
        if plan is not None:
            n_kernel, n_other = hip.graph_node_counts(g.raw_cuda_graph())
            why = None
            if n_other != 0:
                why = f"the captured step holds {n_other} non-kernel node(s) (memset / memcpy)"
            elif n_kernel != plan.nodes:
                why = f"the captured step holds {n_kernel} kernel nodes, {plan.nodes} of them library launches"
            elif plan.streams != 1:
                why = f"the library launches were issued on {plan.streams} streams"
            if why is not None:
                plan.destroy()
                plan, self.why_not_plan = None, why
                if self._sharded:
                    # the capture above holds no collective (the plan had taken them over): once more, as a plain hipGraph
                    del g, out
                    if not had_pool:
                        self._pool = None
                    return self._capture(P, force_graph=True)
            elif self._sharded:
                # (no sections; the static inputs are filled by the staging copy.  Segments tagged for the ahead stream — the
                #  next batch's prepared lookup — run on the sharded path's own plain stream)
                if any(plan.seg_tags):
                    from .sharded import route_stream
                    plan.ahead_stream = route_stream(next(iter(self.X[0].values())).device)
            else:
                # the sections run on the very streams the eager path overlaps on (sort-ahead / first-layer weight gradient):
                # a stream of the plan's own may share the main stream's hardware queue (csrc/plan.hip)
                from . import functional as _Fh
                dev = next(iter(self.X[0].values())).device
                side = _emb._SIDE_STREAMS.get(dev)
                if side is None:
                    side = _emb._SIDE_STREAMS[dev] = hip.make_side_stream(dev)
                side2 = _Fh._WGRAD_STREAMS.get(dev)
                if side2 is None:
                    side2 = _Fh._WGRAD_STREAMS[dev] = hip.make_side_stream(dev, "inline")
                plan.set_streams(side, side2)
                if self.rebind:
                    # an argument that points INSIDE a static input (a derived pointer: an offset view, a column of
                    # a packed buffer) cannot be re-pointed — such a step keeps the staging copy.  (An input no launch reads
                    # at all has no site and needs none; a launch that is not the library's would have refused the plan.)
                    ins = [self.X[P][k] for k in self._keys] + [self.X[1 - P][k] for k in self._keys]
                    sites, interior = plan.bind_report(ins)
                    self.bind_report = {"sites": dict(zip([f"cur:{k}" for k in self._keys] + [f"next:{k}" for k in self._keys], sites)),
                                        "interior": interior}
                    if interior == 0:
                        self._bind_sites[P] = plan.bind_inputs([t.data_ptr() for t in ins])
                    else:
                        self._bind_sites[P] = 0
        self.captures += 1
        lz = getattr(emb, "_lazy", None)
        self._ahead_used = ahead  # (also when the step fell back to a hipGraph: the captured launches are the same)
        self._ahead_noclear[P] = ahead and lz is not None and lz._noclear is not None
        if lz is not None:
            lz._cf_ahead = self._ahead_used
        self._drop_calls[P], self._drop_seed = drop["calls"], drop["seed"]
        self.graphs[P], self.plans[P] = g, plan
        self.backend_used = "plan" if plan is not None else "hipgraph"
        self.outs[P] = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}
        del out

    def _seed_grad(self, loss, value: float = 1.0):
        """the persistent 1.0 a captured backward starts from (created in _alloc, outside any capture); None — the default
        seed, an ATen fill — when the loss is not the float32 scalar every model here returns.  value = 1 / world: the seed of
        a row-sharded step recorded as a launch plan (sharded._SEED_SCALED)"""
        if value != 1.0:
            one = self._seed if (self._seed is not None and self._seed_value == value) else None  # (made in front of the capture)
        else:
            one = self._one
        if one is not None and one.device == loss.device and one.dtype == loss.dtype and one.shape == loss.shape:
            return one
        return None

    def _sharded_capturable(self, batch) -> bool:
        """the fixed-capacity exchange is active for this batch size (measured by an earlier eager step): no split size
        comes back to the host inside the step"""
        emb = self.model.embedding_layer
        n = len(emb.emb_feature) * batch[emb.emb_feature[0]].numel()
        return emb.check_indices == "deferred" and emb._capacity is not None and n <= emb._capacity_n

    def _pinned(self, Q):
        src = tuple(self.X[Q][c] for c in self.model.embedding_layer.emb_feature)
        for c_src, _, c_out in _emb._SORT_PINNED:
            if len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src)):
                return c_out
        raise RuntimeError("GraphedTrainStep: the static batch lost its pinned sort buffers")

    def _lazy_states(self):
        return [st._lazy for st in getattr(self.opt, "_stores", {}).values() if getattr(st, "_lazy", None) is not None]

    def _drop_captures(self):
        self._ahead_drop()
        self._ahead_used = False
        for lz in self._lazy_states():  # (no capture in use rebuilds these states' closed-form windows any more)
            lz._cf_in_capture = False
            lz._cf_ahead = False
        self._bind_sites = [0, 0]
        for pl in self.plans:
            if pl is not None:
                pl.destroy()
        self.plans = [None, None]
        self.graphs, self.outs, self._pool = [None, None], [None, None], None  # (the pool dies with its graphs)

    def reset(self):
        """drop the captured graphs (buffers they hold moved, the model or the batch shape changed)"""
        if torch.cuda.is_available():
            torch.cuda.synchronize()  # never destroy a graph executable that is still in flight
        self._drop_captures()
        if self.X is not None and not self._sharded:
            for x in self.X:
                _emb.EmbeddingLayer.unpin_sorts(x)
        self.X, self._staged = None, None
        self._copy_lists = None

    # ---- the step ------------------------------------------------------------------------------------------------------
    def _fits(self, batch) -> bool:
        x = self.X[0]
        return batch.keys() == x.keys() and all(batch[k].shape == x[k].shape and batch[k].dtype == x[k].dtype and
                                                batch[k].device == x[k].device for k in x)

    # ---- live per-launch timing inside replayed steps (launch plans only) ------------------------------------------------
    def launch_names(self):
        """[(kernel name, section), ...] of the captured step's launches in recorded order (None unless it replays as a plan)"""
        pl = next((p for p in self.plans if p is not None), None)
        return None if pl is None else [pl.launch_name(k) for k in range(pl.nodes)]

    def set_probe(self, launch: int):
        for pl in self.plans:
            if pl is not None:
                pl.set_probe(launch)

    def last_probe_ms(self) -> float:
        return self._last_plan.probe_ms()

    def __call__(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]] = None):
        import time as _time
        t_in = _time.perf_counter()
        w0 = self.host_wait_s
        try:
            return self._call(batch, next_batch)
        finally:
            dt = _time.perf_counter() - t_in
            self.host_call_s += dt
            own = dt - (self.host_wait_s - w0)  # (this call's time without its back-pressure wait)
            if own > self.host_call_max_s:
                self.host_call_max_s = own

    def _call(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]] = None):
        import time as _time
        seg_t = _time.perf_counter()

        def seg(name):  # slowest host time per part of the call (host_seg_max: where a stalled call stalled)
            nonlocal seg_t
            t = _time.perf_counter()
            if t - seg_t > self.host_seg_max.get(name, 0.0):
                self.host_seg_max[name] = t - seg_t
            seg_t = t

        if self.eager_left > 0 or next_batch is None:
            self.eager_left -= 1
            self._staged = None
            return self._eager(batch, next_batch)
        if self.X is None:
            self._alloc(batch)
        if self._sharded and not self._sharded_capturable(batch):
            self._staged = None
            return self._eager(batch, next_batch)
        if not (self._fits(batch) and self._fits(next_batch)):
            # another shape than the captured one (the smaller last batch of an epoch, or the batch before it, whose
            # captured step would sort it): this step runs eagerly
            self._staged = None
            return self._eager(batch, next_batch)
        seg("fits")
        # not the batch announced by the previous call — or its tensors are not the ones (or no longer hold what) that call
        # sorted: stage and sort it now
        if self._staged is not batch or self._staged_sig != self._sig_of(batch):
            self._stage_current(batch)
        P = self.P
        sig = self.opt.prepare_step()
        gen = hip.device_generator(self._drop_clock.device)
        sig = sig + (gen.initial_seed() & 0xFFFFFFFFFFFFFFFF,)  # (the dropout seed is a frozen launch argument: re-seeding re-captures)
        if sig != self._sig:               # a table a graph points into has moved (capacity doubled, replay mode changed)
            if self._sig is not None:
                # launches of the old graphs may still be in flight (the host runs steps ahead): destroying a graph
                # executable under them frees the kernel arguments they read (observed: memory aperture violation)
                torch.cuda.synchronize()
                self._drop_captures()
            self._sig = sig
        lz_a = self._ahead_lazy()
        if lz_a is not None and self._ahead_valid != P and (self.graphs[P] is None or self._ahead_used):
            # the step before this one was not a replay that caught X[P]'s rows up (the first replay, an eager step in
            # between, a restaged batch): the same launch, eagerly, in front of the replay that counts on it
            self._ahead_drop()
            if self.model.embedding_layer.catch_up_ahead(self._pinned(P)[1]):
                self._ahead_valid = P
        if self.graphs[P] is None:
            self._capture(P)
        seg("signature")
        # (staged here, behind the capture: a re-capture a few lines up may have replaced the plan this call started with)
        plan = self.plans[P]
        direct = (self.rebind and plan is not None and self._bind_sites[P] > 0
                  and all(t.is_contiguous() for t in batch.values()) and all(t.is_contiguous() for t in next_batch.values()))
        if direct:
            # the replay reads this batch and sorts the next one straight from the caller's tensors: no staging copy
            plan.set_inputs([batch[k] for k in self._keys] + [next_batch[k] for k in self._keys])
            self._x_valid[1 - P] = False
            self._held.append((batch, next_batch))
            del self._held[:-(self.MAX_IN_FLIGHT + 2)]
        else:
            if not self._x_valid[P]:  # (the step before read this batch from the caller's tensors: X[P] was never filled)
                self._copy(P, batch)
                self._x_valid[P] = True
            self._copy(1 - P, next_batch)
            self._x_valid[1 - P] = True
            if plan is not None and self._bind_sites[P] > 0:
                plan.set_inputs([self.X[P][k] for k in self._keys] + [self.X[1 - P][k] for k in self._keys])
        seg("copy")
        counters = self.opt.host_counters()
        if counters != self._dev:          # eager steps ran in between: bring the device counters to the host's
            self.opt.set_device_clock(True)
            self.opt.set_device_clock(False)
        # bound how far the host runs ahead (a handful of launches in flight is all the overlap there is to win)
        if len(self._inflight) >= self.MAX_IN_FLIGHT:
            import time as _time
            done = self._inflight.pop(0)
            t_w = _time.perf_counter()
            done.synchronize()
            dt_w = _time.perf_counter() - t_w
            self.host_wait_s += dt_w
            self.host_call_s -= dt_w  # (the wait is the device's time, not the host's)
            self._ev_pool.append(done)
            seg_t = _time.perf_counter()
        seg("prepare")
        calls = self._drop_calls[P]
        if calls > 0:
            off = gen.get_offset()
            if self._drop_clock_host != off:  # the first replay, or eager dropout calls since the last one
                self._drop_clock.fill_(off)
            gen.set_offset(off + 4 * calls)   # what `calls` eager dropout launches would have consumed
            self._drop_clock_host = off + 4 * calls
        if self.plans[P] is not None:
            self.plans[P].replay()
            self._last_plan = self.plans[P]
        else:
            self.graphs[P].replay()
        seg("replay")
        ev = self._ev_pool.pop() if self._ev_pool else (torch.cuda.Event() if _TORCH_MARKERS else hip.Marker())
        ev.record()
        self._inflight.append(ev)
        seg("record")
        self.opt.advance_host()
        hip.bump_weight_epoch()
        self._dev = self.opt.host_counters()
        if self._ahead_used:
            # host mirror of what the replay's last launch did: the next batch's rows are caught up and stamped for the step
            # that follows (LazyAdamRows.replay's bookkeeping, which no python ran for)
            emb = self.model.embedding_layer
            lz, sk_next = emb._lazy, self._pinned(1 - P)[1]
            lz._marked_for = lz.t + 1
            lz._noclear = (lz.t + 1, sk_next) if self._ahead_noclear[P] else None
            emb._ahead_done = sk_next
            self._ahead_valid = 1 - P
        seg("advance")
        self.replays += 1
        self.P, self._staged, self._staged_sig = 1 - P, next_batch, self._sig_of(next_batch)
        return self.outs[P]
