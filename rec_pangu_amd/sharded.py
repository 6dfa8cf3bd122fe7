"""Row-sharded embedding tables across the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Nothing like this exists in the reference (single process, single device: SURVEY.md §2.2); its only
contract is "same numbers as the 1-GPU path" (§8e), which tests/test_sharded_gloo.py checks.

Partition.  Global batch B is split into contiguous local batches of b = B/G samples.  Every arena row r
(all tables of the layer back to back, as in EmbeddingLayer) lives on rank r % G at local row r // G:
mod-sharding spreads both capacity and traffic (Criteo's three 5-10 M-row tables hold 76 % of the rows,
its 3..30-row tables are the hottest), without per-table placement decisions.

One exchange each way per step (the only data-path collectives), over the DEDUPLICATED requests of the local batch
(see _Route: ~0.55 M distinct rows out of 1.7 M requests at b = 65536 with Criteo cardinalities):
  forward   ids  : all_to_all_single of int64 local-row ids, bucketed by owner (sizes via a G-int all-to-all)
            rows : owner gathers its rows -> all_to_all_single back ([n_unique, D] fp32; every peer pair has its
                   own xGMI link, so the G-1 transfers of a rank run concurrently)
  backward  grads: request gradients are segment-summed per unique row locally, travel the same route reversed,
                   then a local sort + segmented reduce adds them into the owner's dense gradient
Dense parameters are replicated; `allreduce_dense_grads` averages their gradients in ONE flat bucket
(0.46 MB for the default MLP: latency-bound, so one collective, not one per tensor).
Embedding-row gradients are scaled by 1/G before they travel, so the update equals the 1-GPU update on
the global batch (loss = mean over B).

On a HIP device the local work runs on the HIP kernels (fused gather + FM from the received rows,
sort + segmented reduce for the gradients); on CPU (gloo tests, BASELINE config 0 style) plain torch ops.
"""
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import functional as Fh


_STAGED: dict = {}


def _host_staged(t, group) -> bool:
    """HIP tensors on a gloo group: the collective is staged through host memory.  That is the VIRTUAL multi-rank harness
    (tests/test_hip_sharded_world2.py: two processes sharing one MI355X, where RCCL refuses two ranks on one device) —
    every HIP kernel of the sharded path runs exactly as under RCCL, only the wire is swapped."""
    if not t.is_cuda:
        return False
    k = id(group)
    if k not in _STAGED:
        _STAGED[k] = dist.get_backend(group) == "gloo"
    return _STAGED[k]


def _recorded(fn, t) -> bool:
    """While a LAUNCH PLAN is being recorded (graph_step.GraphedTrainStep on a row-sharded model, round 6) a collective is not
    issued: the plan is cut at this point (rp_plan_host_mark) and every replay issues `fn` itself between the two segments,
    on the replay's stream, with the very buffers of the recording (they live in the capture's pool; the closure keeps them
    referenced).  -> True when the plan took the call over."""
    if not t.is_cuda:
        return False
    from . import hip
    return hip.LaunchPlan.host_call(fn)


def all_to_all(out, inp, out_splits=None, in_splits=None, group=None):
    def issue():
        if _host_staged(inp, group):
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
            out.copy_(h_out)
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
    if not _recorded(issue, inp):
        issue()


def all_reduce(t, op=None, group=None):
    op = dist.ReduceOp.SUM if op is None else op

    def issue():
        if _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
            return
        dist.all_reduce(t, op=op, group=group)
    if not _recorded(issue, t):
        issue()


# A recorded step seeds its backward with 1 / world instead of 1 (graph_step.GraphedTrainStep._seed_grad): every gradient of
# the step — the row gradients that travel AND the dense ones — then carries the 1 / G of "loss = mean over the GLOBAL
# batch" from the start, and neither the row gradients nor the all-reduced bucket need a scaling pass of their own (two ATen
# launches a launch plan must not hold).  [on, world]
_SEED_SCALED = [False]


class seed_scaled:
    """context: the backward inside is seeded with 1 / world (see _SEED_SCALED)"""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _SEED_SCALED[0] = _SEED_SCALED[0], self.on
        return self

    def __exit__(self, *exc):
        _SEED_SCALED[0] = self.prev
        return False


_CONSTS: dict = {}
# the stream the look-ahead of a sharded lookup runs on (route, id exchange, owner-side sort): a PLAIN torch stream of this
# module's own — RCCL's collectives are issued on it, and the lowest-priority ExternalStream the single-device path keeps in
# embedding._SIDE_STREAMS has not run under a process group with more than one rank (sharing that dict made the
# choice depend on which layer created the entry first)
_ROUTE_STREAMS: dict = {}


def route_stream(dev):
    """The lowest-priority stream of hip.make_side_stream, an instance of this module's own (RP_ROUTE_STREAM=plain: a torch
    stream of the normal priority, rounds 3-5 — and the default under an RCCL group of more than one rank, RP_ROUTE_STREAM=low
    overrides).  HIP multiplexes the streams of one priority onto a handful of hardware
    queues, and a plain stream that lands on the main stream's queue runs IN FRONT of or BEHIND the step instead of beside it
    (measured on one MI355X: the recorded sharded step with its look-ahead on a plain stream took exactly the serial time,
    DESIGN.md §8).  Collectives are issued with this stream current (torch's process group orders its own communication
    stream against the CURRENT stream with events, whatever kind of stream that is): covered by the 1-rank RCCL test with
    RP_FORCE_A2A=1 and the two-rank host-staged test; RCCL with N > 1 has not executed on any stream yet."""
    side = _ROUTE_STREAMS.get(dev)
    if side is None:
        # (more than one rank: the plain stream unless asked otherwise — the exchange issued with an ExternalStream current has
        #  only run under a one-rank RCCL group and the host-staged two-rank harness; correctness does not depend on the choice)
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() != "gloo"
        kind = os.environ.get("RP_ROUTE_STREAM", "plain" if multi else "low")
        if kind == "low":
            from . import hip
            side = hip.make_side_stream(dev, "route")
        elif kind == "high":
            side = torch.cuda.Stream(device=dev, priority=-1)
        else:
            side = torch.cuda.Stream(device=dev)
        _ROUTE_STREAMS[dev] = side
    return side


def _const_i64(device, n: int, value: int):
    """a persistent [n] int64 tensor filled with `value` (kernel-argument tables rebuilt every step otherwise: an ATen fill
    launch each, which a launch plan must not hold)"""
    k = (str(device), n, value)
    t = _CONSTS.get(k)
    if t is None:
        if len(_CONSTS) > 64:
            _CONSTS.clear()
        t = _CONSTS[k] = torch.full((n,), value, dtype=torch.int64, device=device)
    return t


def _force_a2a() -> bool:
    """RP_FORCE_A2A=1: a one-rank group still goes through all_to_all_single (RCCL self-copies on HIP tensors), so that
    1-GPU runs exercise the collective call path itself."""
    return os.environ.get("RP_FORCE_A2A", "0") == "1"


def _a2a(inp, out_splits, in_splits, group, world: int):
    """all_to_all_single into a fresh tensor; with one rank the exchange is the identity and `inp` itself comes back (no
    self-copy through an RCCL kernel: 0.12 ms per exchange at Criteo shape) unless RP_FORCE_A2A=1"""
    if world == 1 and not _force_a2a():
        return inp
    out = torch.empty((sum(out_splits),) + tuple(inp.shape[1:]), dtype=inp.dtype, device=inp.device)
    all_to_all(out, inp, out_splits, in_splits, group)
    return out


_WIRE_BYTES = [0]  # payload bytes handed to the row / row-gradient exchanges since the last reset (tests, bench)


def _a2a_rows(rows, out_splits, in_splits, layer):
    """the [n, D] fp32 row (or row-gradient) exchange.  layer.wire_dtype == torch.bfloat16: the rows travel as bf16 (half
    the bytes on every xGMI link: the row exchanges are the wire time of a sharded step) and are widened
    again on arrival — a stated-tolerance mode like the bf16 tables (2^-9 relative per travelling value; logits within
    6e-2), the fp32 wire stays the parity mode.  The bits are exchanged as bytes so that every backend takes them."""
    if layer.world == 1 and not _force_a2a():
        return rows
    if getattr(layer, "wire_dtype", torch.float32) == torch.bfloat16:
        packed = rows.to(torch.bfloat16).view(torch.uint8)  # [n, 2 D] bytes: every backend exchanges uint8
        _WIRE_BYTES[0] += packed.numel()
        return _a2a(packed, out_splits, in_splits, layer.group, layer.world).view(torch.bfloat16).to(torch.float32)
    _WIRE_BYTES[0] += rows.numel() * 4
    return _a2a(rows, out_splits, in_splits, layer.group, layer.world)


class _Route:
    """One lookup exchange: the n = F*b row requests of the local batch are DEDUPLICATED and bucketed by owner.

    A request for global arena row r goes to rank r % G as local row r // G.  Sorting the composite key
    (owner << lbits | local_row) groups the requests by owner and, inside an owner, by row — so the unique rows of
    each owner are contiguous and already in send order, and equal requests are adjacent (one `unique_consecutive`).
    Only the unique rows travel: with Criteo-like cardinalities a 65536-sample batch asks for ~0.55 M distinct rows
    out of 1.7 M requests, which cuts both all-to-alls (rows forward, row gradients backward) by 3x.
      slot_sorted[j] : unique-row slot of the j-th request in sorted order (ascending, runs of equal slots)
      pos_sorted[j]  : which request (pair id p = f*b + i) that is
      slot_of_pair[p]: the slot request p reads its row from
    """

    def __init__(self, keys, layer, pin=None):
        """keys: int64 arena rows [n] (p = f*b + i) — or, on a HIP device, the list of per-field id tensors (the
        range check, the composite keys and everything after the sort then run in rp_shard_keys / rp_route_build).
        pin (HIP, fixed-capacity exchange only): a _PinnedRoute whose persistent buffers the lists are written into."""
        world, lbits = layer.world, layer.lbits
        nbits = lbits + max(1, (world - 1).bit_length())
        if isinstance(keys, (list, tuple)):
            from . import hip
            idx = keys
            n = len(idx) * idx[0].numel()
            err = layer._err_flag(idx[0].device)
            comp = hip.shard_keys(layer._row_base, layer._row_count, idx, world, lbits, err)
            if pin is not None:
                sk, sp = hip.sort_pairs(comp, end_bit=nbits, out=(torch.empty_like(comp), pin.pos_sorted))
                hip._held(sk)
            else:
                sk, sp = hip.sort_pairs(comp, end_bit=nbits)  # stable radix sort (csrc/sort.hip), (key, position) pairs
            self.slot_sorted, self.slot_of_pair, uniq_rows, counts = hip.route_build(
                sk, sp, world, lbits, out=None if pin is None else (pin.slot_sorted, pin.slot_of_pair))
            self.pos_sorted = sp
            self.n_requests = n
            # fixed-capacity exchange (check_indices == "deferred"): every owner gets `cap` slots, the split sizes are
            # constants and nothing comes back to the host.  The capacity was measured on a batch of `_capacity_n`
            # requests; a LARGER batch (an evaluation batch, a changed loader) takes the exact exchange below and the
            # capacity grows.  Every rank must see the same local batch size in a step (the choice is made without a
            # collective).  An owner asked for more than `cap` unique rows by a batch that is not larger sets the error
            # flag, which raise_if_bad_index() — all-reduced over the ranks — turns into an exception on EVERY rank.
            cap = layer._capacity if (layer.check_indices == "deferred" and n <= layer._capacity_n) else None
            if cap is not None:
                self.local_rows = hip.route_pad(sk, sp, world, lbits, cap, counts, self.slot_sorted, self.slot_of_pair, err,
                                                out=None if pin is None else pin.local_rows)
                self.send = self.recv = [cap] * world
                self.n_unique = self.n_recv = cap * world
                self._field_major(sk, sp, n // len(idx), world, lbits, pin)
                return
            if pin is not None:
                raise RuntimeError("_Route: persistent buffers need the fixed-capacity exchange")
            send_counts = counts[:world]
            recv_counts = torch.empty_like(send_counts)
            all_to_all(recv_counts, send_counts, group=layer.group)
            # one host sync per exchange: the collective wants its split sizes on the host
            self.send, self.recv = torch.stack([send_counts, recv_counts]).tolist()
            self.n_unique, self.n_recv = sum(self.send), sum(self.recv)
            self.local_rows = uniq_rows[:self.n_unique]
            self._field_major(sk, sp, n // len(idx), world, lbits)
            if layer.check_indices == "deferred":
                # from the next step on: fixed capacity, 25 % above the largest per-owner count any rank saw now (never
                # below what an earlier measurement gave)
                m = torch.tensor([max(self.send + self.recv), n], dtype=torch.int64, device=sk.device)
                all_reduce(m, op=dist.ReduceOp.MAX, group=layer.group)
                m = m.tolist()
                slack = float(getattr(layer, "capacity_slack", 0.25))
                layer._capacity = max(layer._capacity or 0, (int(m[0] * (1.0 + slack)) + 1024 + 255) // 256 * 256)
                layer._capacity_n = max(layer._capacity_n, m[1])
            return
        comp = ((keys % world) << lbits) | torch.div(keys, world, rounding_mode="floor")
        sk, sp = torch.sort(comp, stable=True)
        uniq, inverse = torch.unique_consecutive(sk, return_inverse=True)
        send_counts = torch.bincount((uniq >> lbits).long(), minlength=world)
        recv_counts = torch.empty_like(send_counts)
        all_to_all(recv_counts, send_counts, group=layer.group)
        self.send, self.recv = torch.stack([send_counts, recv_counts]).tolist()
        self.local_rows = (uniq & ((1 << lbits) - 1)).long().contiguous()
        self.n_unique, self.n_recv = int(uniq.numel()), sum(self.recv)
        self.slot_sorted = inverse.to(torch.int32).contiguous()
        self.pos_sorted = sp.to(torch.int32).contiguous()
        self.slot_of_pair = torch.empty((keys.numel(),), dtype=torch.int64, device=keys.device)
        self.slot_of_pair[sp.long()] = inverse.long()
        self.n_requests = keys.numel()


def _route_field_major(self, sk, sp, b, world, lbits, pin=None):
    """slot_fm / pos_fm: the sorted (slot, position) list in FIELD-major order — what the segment-sum-first backward of the
    fused first layer takes (rp_embed_grad_seg).  One owner: the composite-key order is field-major already."""
    if world == 1:
        self.slot_fm, self.pos_fm = self.slot_sorted, self.pos_sorted
        return
    from . import hip
    self.slot_fm, self.pos_fm = hip.route_field_major(sk, sp, self.slot_sorted, b, world, lbits,
                                                      out=None if pin is None else (pin.slot_fm, pin.pos_fm))


_Route._field_major = _route_field_major
_Route.slot_fm = _Route.pos_fm = None  # (routes built with torch ops — CPU — have no field-major list)


class _PinnedRoute:
    """The prepared lookup of ONE static batch of a recorded step (graph_step.GraphedTrainStep, round 6) in PERSISTENT
    buffers: its route, the rows this rank is asked for and their owner-side sort.  A replayed step reads them; the replay
    of the step BEFORE it (or, for the first replay / a restaged batch, an eager pass) fills them — on the ahead stream,
    beside the step (ShardedEmbeddingLayer.route_into).  Looks like a _Route to _ShardedRows / _RowsToLinear."""

    def __init__(self, layer, n: int, b: int):
        dev = layer.local_arena.device
        world, cap = layer.world, layer._capacity
        i32 = lambda m: torch.zeros((m,), dtype=torch.int32, device=dev)
        i64 = lambda m: torch.zeros((m,), dtype=torch.int64, device=dev)
        self.slot_sorted, self.pos_sorted, self.slot_of_pair = i32(n), i32(n), i64(n)
        self.slot_fm, self.pos_fm = (self.slot_sorted, self.pos_sorted) if world == 1 else (i32(n), i32(n))
        self.local_rows = i64(world * cap)
        self.recv_rows = self.local_rows if (world == 1 and not _force_a2a()) else i64(world * cap)
        self.served = (i32(world * cap), i32(world * cap))
        self.send = self.recv = [cap] * world
        self.n_unique = self.n_recv = cap * world
        self.n_requests, self.capacity, self.b = n, cap, b


class _ShardedRows(torch.autograd.Function):
    """rows[s] = global_arena[unique request s]; backward routes the per-unique-row gradients back to the owners
    and reduces them into the owner's local dense gradient."""

    @staticmethod
    def forward(ctx, layer, keys, local_arena):
        dev = local_arena.device
        prepared, layer._prepared = layer._prepared, None
        # (prepared: what was done ahead on the side stream for this batch — its route (ShardedEmbeddingLayer._route_ahead)
        #  and, once the previous step's collectives were enqueued, its id exchange and owner-side sort (_request_ahead))
        route, recv_rows, served_sorted = prepared if prepared is not None else (None, None, None)
        if route is None:
            route = _Route(keys, layer)
        if recv_rows is None:
            recv_rows = _a2a(route.local_rows, route.recv, route.send, layer.group, layer.world)
        # (grad mode is off inside a Function.forward: the deferred optimizer is told explicitly whether a backward follows)
        served = layer._local_gather(recv_rows, served_sorted, mark=bool(ctx.needs_input_grad[2]))  # [n_recv, D]
        ctx.presorted = getattr(layer, "_served_sorted", None)
        layer._served_sorted = None
        rows = _a2a_rows(served, route.send, route.recv, layer)
        ctx.layer, ctx.route = layer, route
        ctx.scaled, layer._scaled = layer._scaled, None
        ctx.save_for_backward(recv_rows)
        ctx.mark_non_differentiable(route.slot_of_pair, route.slot_sorted, route.pos_sorted)
        # (autograd materialised a zero "gradient" for each of the three index outputs: three ATen fill launches per step)
        ctx.set_materialize_grads(False)
        layer._route_fm = (route.slot_fm, route.pos_fm)  # (read by gather_linear right after this call: not autograd outputs)
        return rows, route.slot_of_pair, route.slot_sorted, route.pos_sorted

    @staticmethod
    def backward(ctx, g_rows, *_unused):
        (recv_rows,) = ctx.saved_tensors
        layer, route = ctx.layer, ctx.route
        if g_rows is None:  # (nothing downstream asked for the rows' gradient)
            return None, None, None
        # 1/G: the update must equal the 1-GPU update on the global batch (unless the producer already folded it in)
        prescaled = layer.world == 1 or (ctx.scaled is not None and ctx.scaled[0]) or _SEED_SCALED[0]
        g_rows = g_rows.contiguous() if prescaled else (g_rows * (1.0 / layer.world)).contiguous()
        recv_g = _a2a_rows(g_rows, route.recv, route.send, layer)
        layer._local_scatter_add(recv_rows, recv_g, presorted=ctx.presorted)
        return None, None, None


class _RowsToX(torch.autograd.Function):
    """HIP: x[b, ldx] (+ FM) from the unique rows received (request p reads slot slot_of_pair[p]); the backward is
    the segmented sum of the request gradients per unique row (the same kernel the single-GPU gather backward
    uses), so what travels back is again one row per unique request."""

    @staticmethod
    def forward(ctx, rows, slot_of_pair, dense: List[torch.Tensor], slot_sorted, pos_sorted, b: int, F: int, ldx: int,
                want_fm: bool, err_flag):
        from . import hip
        n, D = rows.shape
        dev = rows.device
        zero, cnt = _const_i64(dev, F, 0), _const_i64(dev, F, n)
        idx = [slot_of_pair[f * b:(f + 1) * b] for f in range(F)]
        x, fm, ssum, _ = hip.embed_gather_fwd(rows, zero, cnt, idx, dense, ldx, want_fm, want_fm, False, err_flag)
        ctx.cfg = (b, F, D, want_fm)
        ctx.save_for_backward(rows, slot_sorted, pos_sorted, ssum)
        return (x, fm) if want_fm else x

    @staticmethod
    def backward(ctx, dx, dfm=None):
        from . import hip
        rows, slot_sorted, pos_sorted, ssum = ctx.saved_tensors
        b, F, D, want_fm = ctx.cfg
        # zero-initialised: slots no request reads (fixed-capacity padding) carry a zero gradient
        g_rows = hip.zeros(rows.shape, rows.dtype, rows.device)
        gfm = dfm.contiguous() if (want_fm and dfm is not None) else None
        dx = None if dx is None else Fh._unit_inner(dx)
        hip.embed_grad_reduce(slot_sorted, pos_sorted, b, D, dx, gfm, ssum if gfm is not None else None,
                              rows if gfm is not None else None, g_rows, accumulate=False)
        return g_rows, None, None, None, None, None, None, None, None, None


class _RowsToLinear(torch.autograd.Function):
    """HIP: _RowsToX fused with the 64-wide Linear + ReLU that consumes x (DeepFM's dnn.net.0), as on the single-GPU
    path: forward = rp_embed_gather_linear_fwd with the received unique rows as the arena; backward = the weight
    gradient and rp_embed_grad_gemm (the layer's dgrad formed inside the per-unique-row segmented sum: dX never
    exists), so what travels back is again one gradient row per unique request."""

    @staticmethod
    def forward(ctx, rows, slot_of_pair, dense: List[torch.Tensor], slot_sorted, pos_sorted, b: int, F: int, ldx: int,
                weight, bias, out_link, err_flag, scale: float, scaled, fm_lists=None):
        from . import hip
        n, D = rows.shape
        dev = rows.device
        zero, cnt = _const_i64(dev, F, 0), _const_i64(dev, F, n)
        idx = [slot_of_pair[f * b:(f + 1) * b] for f in range(F)]
        K = weight.shape[1]
        # round 6, as on the single-device path (functional._EmbedGatherLinear "seg"): NO activation is stored — the embedding
        # columns of the layer's weight gradient come out of the gather backward (rp_embed_grad_seg over the received rows, the
        # route's FIELD-major pair list), the dense columns from a small weight gradient over xd [b, 64].  RP_GRAD_SEG=0: the
        # stored x + rp_linear_wgrad + rp_embed_grad_gemm of rounds 3-5.
        seg = (fm_lists is not None and fm_lists[0] is not None and K > F * D and F <= 64 and D == 64 and weight.shape[0] == 64
               and (ctx.needs_input_grad[8] or ctx.needs_input_grad[0]) and os.environ.get("RP_GRAD_SEG", "1") != "0")
        x, h1, fm, ssum, _ = hip.embed_gather_linear_fwd(rows, zero, cnt, idx, dense, ldx, Fh._rows16(weight), bias, True,
                                                         True, False, err_flag, x_mode="dense" if seg else "full")
        ctx.cfg = (b, D, K, bias is not None, out_link, scale, scaled, seg, F)
        if seg:
            slot_sorted, pos_sorted = fm_lists
        ctx.save_for_backward(rows, slot_sorted, pos_sorted, ssum, x, h1, weight)
        return h1, fm

    @staticmethod
    def backward(ctx, dh1, dfm):
        from . import hip
        rows, slot_sorted, pos_sorted, ssum, x, h1, weight = ctx.saved_tensors
        b, D, K, has_bias, lk, scale, scaled, seg, F = ctx.cfg
        dh1 = Fh._unit_inner(dh1)
        masked = lk is not None and lk.dx is not None and lk.dx.data_ptr() == dh1.data_ptr() and lk.dx.shape == dh1.shape
        if lk is not None:
            lk.dx = None
        dpre = dh1 if masked else hip.relu_bwd(dh1, h1)
        dw = db = None
        if seg:
            # x = xd [b, 64] (the dense columns); slot_sorted / pos_sorted = the route's field-major lists.  dw's dense columns
            # and the bias gradient from the UNSCALED dpre (like every dense gradient: the all-reduce carries their 1/G, or the
            # recorded step's seed), the embedding columns from rp_embed_grad_seg — scaled with the row gradients when the 1/G
            # is folded into dpre here, and scaled back below in that (eager, world > 1) case
            Kg = F * D
            dw = torch.empty((64, K), dtype=torch.float32, device=dpre.device)
            _, db = hip.linear_wgrad(dpre, x, K - Kg, dw=dw[:, Kg:], want_bias=has_bias)
            gfm = dfm.contiguous() if dfm is not None else None
            fold = scale != 1.0 and not _SEED_SCALED[0]
            dpre_r, gfm_r = dpre, gfm
            if fold:
                dpre_r = dpre * scale
                gfm_r = None if gfm is None else gfm * scale
                scaled[0] = True
            g_rows = hip.zeros(rows.shape, rows.dtype, rows.device)  # slots no request reads (padding): a zero gradient
            hip.embed_grad_seg(slot_sorted, pos_sorted, b, D, dpre_r, weight, gfm_r, ssum if gfm_r is not None else None, rows,
                               g_rows, accumulate=False, dw=dw)
            if fold:
                dw[:, :Kg].mul_(1.0 / scale)
            if not ctx.needs_input_grad[0]:
                g_rows = None
            return g_rows, None, None, None, None, None, None, None, dw, db, None, None, None, None, None
        if ctx.needs_input_grad[8] or (has_bias and ctx.needs_input_grad[9]):
            dw, db = hip.linear_wgrad(dpre, x, K, want_bias=has_bias)
        g_rows = None
        if ctx.needs_input_grad[0]:
            wt = hip.transpose(weight, rows_out=x.shape[1])
            gfm = dfm.contiguous() if dfm is not None else None
            if scale != 1.0 and not _SEED_SCALED[0]:
                # the 1/G of the row gradients that travel, folded into the two small operands the rows are linear in
                # (instead of a pass over [n_unique, D] afterwards; a recorded step's backward is seeded with it: _SEED_SCALED)
                dpre = dpre * scale
                gfm = None if gfm is None else gfm * scale
                scaled[0] = True
            g_rows = hip.zeros(rows.shape, rows.dtype, rows.device)  # slots no request reads (padding): a zero gradient
            hip.embed_grad_gemm(slot_sorted, pos_sorted, b, D, dpre, wt, None, gfm, ssum if gfm is not None else None,
                                rows, g_rows, accumulate=False)
        return g_rows, None, None, None, None, None, None, None, dw, db, None, None, None, None, None


class ShardedEmbeddingLayer(nn.Module):
    """Drop-in for EmbeddingLayer inside a model whose tables are row-sharded over `world` ranks."""

    def __init__(self, full_layer, world: int, rank: int, group=None):
        """Cut this rank's shard out of an existing (fully built) EmbeddingLayer — tests and small models.  Large models
        are built sharded from the start (`from_spec` via `build_sharded_model`): no rank ever holds the full arena."""
        super().__init__()
        full_layer._ensure_packed()
        rows = [p.shape[0] for p in full_layer.table_parameters()]
        self._setup(full_layer.enc_dict, full_layer.embedding_dim, list(full_layer.emb_feature), rows, world, rank, group,
                    full_layer.arena.device)
        self.check_indices = full_layer.check_indices
        # rows r with r % world == rank, in order: local row = r // world
        self.local_arena = nn.Parameter(full_layer.arena.detach()[rank::world].clone())
        self._tag()

    @classmethod
    def from_spec(cls, enc_dict, embedding_dim: int, world: int, rank: int, group=None, device=None):
        """Build ONLY this rank's shard, consuming the global RNG exactly as EmbeddingLayer.__init__ does (one
        N(0,1) draw per table, in enc_dict order): every table is drawn into a temporary of its full shape and the
        rows r % world == rank are kept, so peak memory is the shard plus ONE table, and the values are those of the
        single-process construction."""
        self = cls.__new__(cls)
        nn.Module.__init__(self)
        cols = [c for c in enc_dict.keys() if "vocab_size" in enc_dict[c].keys()]
        rows = [enc_dict[c]["vocab_size"] + 1 for c in cols]
        dev = torch.empty(0, device=device).device  # resolves `with torch.device(...)` defaults
        self._setup(enc_dict, embedding_dim, cols, rows, world, rank, group, dev)
        n_local = (self.total_rows - rank + world - 1) // world
        self.local_arena = nn.Parameter(torch.empty((n_local, embedding_dim), device=dev))
        self._tag()
        self.init_tables(nn.init.normal_)
        return self

    def _setup(self, enc_dict, embedding_dim, emb_feature, rows, world, rank, group, device):
        self.enc_dict = enc_dict
        self.embedding_dim = embedding_dim
        self.emb_feature = emb_feature
        self.world, self.rank, self.group = world, rank, group
        self.check_indices = "sync"
        self.wire_dtype = torch.float32  # torch.bfloat16: rows and row gradients travel as bf16 (_a2a_rows)
        base = [0]
        for r in rows[:-1]:
            base.append(base[-1] + r)
        self._rows, self._base = list(rows), base
        self.total_rows = sum(rows)
        self.lbits = max(1, int((self.total_rows + world - 1) // world).bit_length())  # bits of a local row id
        self.register_buffer("_row_base", torch.tensor(base, dtype=torch.int64, device=device), persistent=False)
        self.register_buffer("_row_count", torch.tensor(rows, dtype=torch.int64, device=device), persistent=False)
        self._touched, self._touched_unsorted = None, False
        self._grad_buf = None
        self._lazy = None
        self._served_sorted = None  # (sorted local rows, positions) of the requests being served, reused in backward
        self._err = None
        # fixed-capacity exchange: per-owner slots = (1 + capacity_slack) x the largest per-owner count any rank saw on the
        # measuring batch (+ 1024, rounded to 256).  0.25 is the safe default for drifting id distributions; every slot
        # above the real counts is padding on the wire (a stationary stream holds its counts to a fraction of a per cent:
        # bench.py runs with 1/16)
        self.capacity_slack = 0.25
        self._capacity = None  # per-owner slots of the fixed-capacity exchange (check_indices == "deferred", HIP)
        self._capacity_n = 0   # requests of the (largest) batch the capacity was measured on
        self._prepared = None  # (route, requested rows, their sort) of the batch about to be looked up, as far as prepared ahead
        # the owner-side gradient launch (rp_embed_grad_reduce over the sorted rows this rank was asked for) writes EVERY row of
        # its key list — the list the catch-up launch in front of the lookup stamped: that launch may leave the gradient rows it
        # applied uncleared (1 of its 8 row transfers; LazyAdamRows.replay, as on the single-device path)
        self._noclear_ok = True
        self._scaled = None     # per-lookup flag shared by _ShardedRows and _RowsToLinear (who applies the 1/G)
        self._announced = None  # the batch of the next step (prefetch_sort), until its route is started
        self._ahead = None      # (id tensors, versions, route, event[, requested rows, their sort]) prepared on the side stream
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_lazy())

    def _tag(self):
        ref = weakref.ref(self)
        self.local_arena._rp_store = ref       # FusedAdam finds the arena (and its lazy state) through this
        self.local_arena._rp_shard_init = ref  # BaseModel.reset_parameters draws the tables through init_tables()

    def _table_slice(self, f: int):
        """(first row of table f owned by this rank, its local row) — the rank owns every world-th row from there"""
        first = (self.rank - self._base[f]) % self.world
        return first, (self._base[f] + first) // self.world

    @torch.no_grad()
    def init_tables(self, init_fn):
        """Apply `init_fn` (nn.init.normal_ / kaiming_normal_ / xavier_normal_: the reference's schemes) table by
        table in field order, as it would run over the F full [V+1, D] tables, keeping this rank's rows."""
        self.flush_lazy()
        a = self.local_arena.data
        for f, r in enumerate(self._rows):
            tmp = torch.empty((r, self.embedding_dim), dtype=a.dtype, device=a.device)
            init_fn(tmp)
            first, lrow = self._table_slice(f)
            mine = tmp[first::self.world]
            a[lrow:lrow + mine.shape[0]].copy_(mine)
            del tmp

    # ---- EmbeddingLayer-compatible surface ------------------------------------------------------
    @property
    def arena(self):
        return self.local_arena

    # ---- the "store" protocol FusedAdam / LazyAdamRows use (same as EmbeddingLayer) ------------------------
    @property
    def grad_arena(self):
        return self._grad_buf

    def table_parameters(self):
        return [self.local_arena]

    def _meta(self):
        return (None, None, None, max(1, int(self.local_arena.shape[0] - 1).bit_length()))

    def grads_are_arena(self) -> bool:
        g = self.local_arena.grad
        return g is not None and self._grad_buf is not None and g.data_ptr() == self._grad_buf.data_ptr()

    def grads_were_zeroed(self):
        self._touched, self._touched_unsorted = None, False

    @property
    def _grad_clean(self) -> bool:
        """the next owner-side gradient launch overwrites (no backward has written since the last zero_grad / fused step)"""
        return self._grad_buf is not None and self._touched is None

    def flush_lazy(self):
        if self._lazy is not None:
            self._lazy.flush(self)

    def raise_if_bad_index(self):
        """Turn the device-side flag into the reference's IndexError (bit 0: id out of range) or a RuntimeError (bit 1: the
        fixed-capacity exchange overflowed).  With more than one rank the flag is OR-ed over the ranks first, so that
        every rank raises — and re-measures its capacity — together: a rank that kept going alone would meet its peers
        in mismatched collectives.  COLLECTIVE for world > 1: every rank must call it at the same point (the forward does
        in 'sync' mode, the trainer / bench after the epoch in 'deferred' mode)."""
        if self.world > 1 and dist.is_available() and dist.is_initialized():
            self._or_flag(self._err_flag(self.local_arena.device))
        if self._err is not None:
            code = int(self._err.item())
            if code != 0:
                self._err.zero_()
                if code & 2:
                    cap, self._capacity, self._capacity_n = self._capacity, None, 0  # re-measured by the next (exact) exchange
                    raise RuntimeError(f"row exchange: an owner was asked for more than the fixed capacity of {cap} "
                                       "unique rows; the steps since the last check dropped requests")
                raise IndexError("index out of range in self")

    def _or_flag(self, flag):
        """bitwise OR of the int32 flag over the ranks: ONE MAX all-reduce over its two bits (every backend has MAX)"""
        bits = torch.stack([flag & 1, (flag >> 1) & 1]).reshape(-1)
        all_reduce(bits, op=dist.ReduceOp.MAX, group=self.group)
        flag.copy_((bits[0] | (bits[1] << 1)).reshape(1))

    def _err_flag(self, device):
        if self._err is None or self._err.device != device:
            self._err = torch.zeros((1,), dtype=torch.int32, device=device)
        return self._err

    def _requests(self, X):
        """what _Route takes: the per-field id tensors on a HIP device whose composite keys fit an int32 (the kernels
        do the rest), else the int64 arena-row keys built with torch ops."""
        if self.local_arena.is_cuda and self.lbits + max(1, (self.world - 1).bit_length()) <= 31:
            return [X[c].long().reshape(-1).contiguous() for c in self.emb_feature]
        return self._keys(X)

    def _keys(self, X):
        idx = torch.stack([X[c].long().reshape(-1) for c in self.emb_feature])  # [F, b]
        bad = (idx < 0) | (idx >= self._row_count[:, None])
        self._err_flag(idx.device)
        self._err |= bad.any().to(torch.int32)
        # like the kernel: flag, then use row 0 so the exchange itself stays well-formed on every rank
        idx = torch.where(bad, torch.zeros_like(idx), idx)
        return (idx + self._row_base[:, None]).reshape(-1)  # p = f*b + i

    # ---- the part of a lookup that does not depend on the weights, started ahead -----------------------------
    def prefetch_sort(self, X) -> None:
        """Announce the batch of the NEXT step (as EmbeddingLayer.prefetch_sort).  With the fixed-capacity exchange active
        (HIP, check_indices == 'deferred': nothing comes back to the host) its route — range check, composite keys, sort,
        dedup, padding: nothing that depends on the weights — is built on the side stream behind this layer's next
        forward.  The collectives stay on the caller's stream: RCCL runs them in issue order, and an id exchange queued
        behind the side stream's sort would hold up the gradient exchange of the step in flight."""
        if self.local_arena.is_cuda and all(X[c].device == self.local_arena.device for c in self.emb_feature):
            self._announced = X

    def _route_ahead(self) -> None:
        X, self._announced = self._announced, None
        if X is None:
            return
        if self._capacity is None or self.check_indices != "deferred" \
                or self.lbits + max(1, (self.world - 1).bit_length()) > 31 \
                or len(self.emb_feature) * X[self.emb_feature[0]].numel() > self._capacity_n:
            return  # (a batch larger than the one the capacity was measured on takes the exact exchange, in its own step)
        dev = self.local_arena.device
        side = route_stream(dev)
        src = tuple(X[c] for c in self.emb_feature)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            prepared = _Route(self._requests(X), self)
            event = torch.cuda.Event()
            event.record(side)
        for t in src:
            t.record_stream(side)
        self._ahead = (src, tuple(t._version for t in src), prepared, event)

    def _take_prepared(self, X) -> None:
        """the forward of batch X begins: use what _route_ahead prepared for exactly these tensors, if anything"""
        a, self._ahead, self._prepared = self._ahead, None, None
        pin, self._pinned_next = getattr(self, "_pinned_next", None), None
        if pin is not None:
            self._prepared = (pin, pin.recv_rows, pin.served if self._lazy is not None else None)
            return
        if a is None:
            return
        src, ver, prepared, event = a[:4]
        cur = tuple(X[c] for c in self.emb_feature)
        if len(cur) != len(src) or any(x is not y for x, y in zip(cur, src)) or ver != tuple(t._version for t in cur) \
                or self._capacity is None:
            return
        stream = torch.cuda.current_stream(self.local_arena.device)
        stream.wait_event(event)
        recv_rows, served_sorted = a[4:] if len(a) == 6 else (None, None)
        for t in (prepared.slot_sorted, prepared.slot_of_pair, prepared.pos_sorted, prepared.local_rows, recv_rows) \
                + tuple(served_sorted or ()):
            if t is not None:
                t.record_stream(stream)
        self._prepared = (prepared, recv_rows, served_sorted)

    # ---- the prepared lookup of a recorded step's static batch, in persistent buffers (graph_step, round 6) ---------------
    def pinned_route(self, X):
        """persistent buffers for the prepared lookup of a batch shaped like X (None: the fixed-capacity exchange is not
        active, or the batch is larger than the one its capacity was measured on)"""
        b = X[self.emb_feature[0]].numel()
        n = len(self.emb_feature) * b
        if self._capacity is None or self.check_indices != "deferred" or n > self._capacity_n or not self.local_arena.is_cuda \
                or self.lbits + max(1, (self.world - 1).bit_length()) > 31:
            return None
        return _PinnedRoute(self, n, b)

    def route_into(self, X, pin) -> None:
        """Everything of X's lookup that does not depend on the weights — route, id exchange, owner-side sort of the rows
        asked for — into `pin` (library launches + one collective; under a plan recording the collective is a host mark).
        The results are computed into fresh buffers and moved with ONE launch (rp_multi_copy)."""
        from . import hip
        if pin.capacity != self._capacity:
            raise RuntimeError("route_into: the exchange capacity changed since the buffers were made")
        n = len(self.emb_feature) * X[self.emb_feature[0]].numel()
        if n != pin.n_requests or n > self._capacity_n:
            raise RuntimeError("route_into: the batch does not take the fixed-capacity exchange these buffers were made for")
        route = _Route(self._requests(X), self, pin=pin)  # (every list straight into its persistent buffer)
        if pin.recv_rows is not pin.local_rows:
            all_to_all(pin.recv_rows, route.local_rows, route.recv, route.send, self.group)
        if self._lazy is not None:
            hip.sort_pairs(self._rows_i32(pin.recv_rows), end_bit=self._meta()[3], out=pin.served)

    def use_pinned(self, pin) -> None:
        """the NEXT lookup of this layer reads its prepared part from `pin` (graph_step sets it for the recording of a step and
        for nothing else: a replay runs no python)"""
        self._pinned_next = pin

    def _request_ahead(self) -> None:
        """Second half of the look-ahead, called once the current step's collectives are enqueued (allreduce_dense_grads):
        the id exchange of the announced batch and the owner-side sort of the rows this rank is asked for, on the side
        stream behind the route.  Issued HERE, not with the route: RCCL runs collectives in issue order, and an id exchange
        waiting for the side stream must not sit in front of the gradient exchange / all-reduce of the step in flight."""
        a = self._ahead
        if a is None or len(a) != 4 or self._capacity is None:
            return
        src, ver, route, _ = a
        side = route_stream(self.local_arena.device)
        with torch.cuda.stream(side):
            recv_rows = _a2a(route.local_rows, route.recv, route.send, self.group, self.world)
            served_sorted = None
            if self._lazy is not None and recv_rows.numel():
                from . import hip
                served_sorted = hip.sort_pairs(self._rows_i32(recv_rows), end_bit=self._meta()[3])
            event = torch.cuda.Event()
            event.record(side)
        self._ahead = (src, ver, route, event, recv_rows, served_sorted)

    # ---- local primitives: HIP kernels on a HIP device, torch ops on CPU ---------------------------
    def _local_gather(self, rows_idx, served_sorted=None, mark=None):
        if self.local_arena.is_cuda:
            from . import hip
            n = rows_idx.numel()
            if n == 0:
                return torch.empty((0, self.embedding_dim), device=rows_idx.device)
            self._served_sorted = None
            if self._lazy is not None and self._lazy.t > 0:
                # exact lazy dense Adam: rows about to be served first replay the steps they skipped
                sk, sp = served_sorted if served_sorted is not None else \
                    hip.sort_pairs(self._rows_i32(rows_idx), end_bit=self._meta()[3])
                self._lazy.replay(self, sk, mark=mark)
                self._served_sorted = (sk, sp)
            zero, cnt = _const_i64(rows_idx.device, 1, 0), _const_i64(rows_idx.device, 1, self.local_arena.shape[0])
            x, _, _, _ = hip.embed_gather_fwd(self.local_arena.detach(), zero, cnt, [rows_idx], [], self.embedding_dim,
                                              False, False, False, self._err)
            return x
        return self.local_arena.detach()[rows_idx]

    def _rows_i32(self, rows_idx):
        """the requested local rows (int64, as they travelled) as the int32 keys the row sort takes: rp_embed_keys with one
        table of local_arena.shape[0] rows at base 0 — the cast as a library launch (`.to(torch.int32)` is an ATen one)"""
        from . import hip
        dev = rows_idx.device
        return hip.embed_keys(_const_i64(dev, 1, 0), _const_i64(dev, 1, self.local_arena.shape[0]), [rows_idx],
                              self._err_flag(dev))

    def _local_scatter_add(self, rows_idx, g, presorted=None):
        p = self.local_arena
        if p.is_cuda:
            from . import hip
            fresh = not self.grads_are_arena()
            if self._grad_buf is None or self._grad_buf.shape != p.shape or self._grad_buf.device != p.device:
                self._grad_buf = torch.zeros_like(p)
                self._touched, self._touched_unsorted = None, False
            elif fresh and self._touched is not None:
                hip.zero_rows(self._touched, self.embedding_dim, self._grad_buf)  # zero_grad() dropped the old rows
                self._touched, self._touched_unsorted = None, False
            clean = self._touched is None
            if rows_idx.numel():
                if presorted is not None:
                    sk, sp = presorted
                else:
                    sk, sp = hip.sort_pairs(self._rows_i32(rows_idx), end_bit=self._meta()[3])
                if self._lazy is not None and self._lazy._noclear is not None:
                    # the catch-up in front of this step's lookup left its applied gradient rows uncleared, counting on THIS
                    # launch to overwrite them: kept if it writes that very key list without accumulating
                    self._lazy.resolve_noclear(self, sk if clean else None)
                hip.embed_grad_reduce(sk, sp, rows_idx.numel(), self.embedding_dim, g, None, None, None,
                                      self._grad_buf, accumulate=not clean)
                if self._touched is None:
                    self._touched, self._touched_unsorted = sk, False
                else:
                    self._touched, self._touched_unsorted = torch.cat([self._touched, sk]), True
            if fresh:
                p.grad = self._grad_buf
        else:
            upd = torch.zeros_like(p).index_add_(0, rows_idx, g)
            p.grad = upd if p.grad is None else p.grad + upd

    # ---- forward ------------------------------------------------------------------------------------
    def gather_concat(self, X, dense: List[torch.Tensor], want_fm: bool, pad_to: int = 64):
        keys = self._requests(X)
        F, D = len(self.emb_feature), self.embedding_dim
        b = keys[0].numel() if isinstance(keys, list) else keys.numel() // F
        self._take_prepared(X)
        rows, slot_of_pair, slot_sorted, pos_sorted = _ShardedRows.apply(self, keys, self.local_arena)
        d = F * D + len(dense)
        ldx = (d + pad_to - 1) // pad_to * pad_to
        dense = [t.float().reshape(-1).contiguous() for t in dense]
        out = _RowsToX.apply(rows, slot_of_pair, dense, slot_sorted, pos_sorted, b, F, ldx, want_fm, self._err)
        self._route_ahead()
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return out if want_fm else (out, None)

    def gather_linear_fits(self, n_dense: int, linear: nn.Linear, pad_to: int = 64) -> bool:
        """as EmbeddingLayer.gather_linear_fits: can rows->x + FM + `linear` (+ ReLU) run as one fused launch?"""
        from . import hip
        F, D = len(self.emb_feature), self.embedding_dim
        d = F * D + n_dense
        ldx = (d + pad_to - 1) // pad_to * pad_to
        return (self.local_arena.is_cuda and D == 64 and linear.out_features == 64 and linear.in_features == d
                and self.lbits + max(1, (self.world - 1).bit_length()) <= 31
                and hip.get_matmul_precision() != "fp32"
                and hip.embed_gather_linear_fits(D, F, n_dense, 64, ldx, Fh._rows16(linear.weight)))

    def gather_linear(self, X, dense: List[torch.Tensor], linear: nn.Linear, out_link, pad_to: int = 64):
        """(h1 [b, 64] = relu(linear(cat(emb, dense))), fm [b, 1]) from the exchanged unique rows, one fused launch"""
        keys = self._requests(X)
        F, D = len(self.emb_feature), self.embedding_dim
        b = keys[0].numel()
        self._take_prepared(X)
        scaled = self._scaled = [False]  # set by _RowsToLinear.backward once it has applied the 1/G itself
        rows, slot_of_pair, slot_sorted, pos_sorted = _ShardedRows.apply(self, keys, self.local_arena)
        d = F * D + len(dense)
        ldx = (d + pad_to - 1) // pad_to * pad_to
        dense = [t.float().reshape(-1).contiguous() for t in dense]
        fm_lists, self._route_fm = getattr(self, "_route_fm", None), None
        out = _RowsToLinear.apply(rows, slot_of_pair, dense, slot_sorted, pos_sorted, b, F, ldx, linear.weight, linear.bias,
                                  out_link, self._err, 1.0 / self.world, scaled, fm_lists)
        self._route_ahead()
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return out

    def forward(self, X: Dict[str, torch.Tensor], name: Optional[str] = None) -> torch.Tensor:
        if name is not None:
            raise NotImplementedError("by-name lookups are not sharded (off the ranking hot path)")
        F, D = len(self.emb_feature), self.embedding_dim
        if self.local_arena.is_cuda:
            x, _ = self.gather_concat(X, [], want_fm=False, pad_to=1)
            return x.view(x.shape[0], F, D)
        keys = self._keys(X)
        rows, slot_of_pair, _, _ = _ShardedRows.apply(self, keys, self.local_arena)
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return rows[slot_of_pair].view(F, -1, D).permute(1, 0, 2)

    # ---- checkpoints in the reference layout ------------------------------------------------------------
    def full_tables(self) -> Dict[str, torch.Tensor]:
        """All-gather the shards and cut them back into the reference's per-table tensors
        (`embedding_layer.<col>.weight`), e.g. for RankTrainer.save_all on rank 0."""
        self.flush_lazy()
        per = (self.total_rows + self.world - 1) // self.world
        mine = torch.zeros((per, self.embedding_dim), dtype=self.local_arena.dtype, device=self.local_arena.device)
        mine[:self.local_arena.shape[0]] = self.local_arena.detach()
        if _host_staged(mine, self.group):
            mine = mine.cpu()
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        full = torch.stack(parts, dim=1).reshape(per * self.world, self.embedding_dim)[:self.total_rows]
        out, off = {}, 0
        for col, r in zip(self.emb_feature, self._row_count.tolist()):
            out[col] = full[off:off + r].clone()
            off += r
        return out


class _SyncBatchNormFn(torch.autograd.Function):
    """BatchNorm1d over the GLOBAL batch (all ranks' local batches).  Forward: all-reduce of (sum x, count), then of
    sum (x - mean)^2 (variance around the GLOBAL mean: no E[x^2] - E[x]^2 cancellation); backward: one all-reduce of
    (sum dy, sum dy * xhat).  On a HIP device the column sums and the element-wise passes are the rp_batchnorm_*
    kernels (two-stage deterministic reductions), on the CPU torch ops.  Gradients follow the local-loss convention of
    this module (every rank backpropagates its own mean loss; row gradients are scaled 1/G when they travel and dense
    gradients are averaged by allreduce_dense_grads), so the means below are over all N = sum of local batch sizes."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float, group):
        C = x.shape[1]
        hipdev = x.is_cuda
        if hipdev:
            from . import hip
            x = Fh._unit_inner(x)
            s1 = hip.batchnorm_colsum(x)
        else:
            s1 = x.sum(0)
        stats = torch.cat([s1, x.new_tensor([float(x.shape[0])])])
        all_reduce(stats, group=group)
        n = stats[-1]
        mean = stats[:C] / n
        s2 = hip.batchnorm_colsum(x, mean.contiguous()) if hipdev else ((x - mean) ** 2).sum(0)
        s2 = s2.clone()
        all_reduce(s2, group=group)
        var = s2 / n  # biased, as F.batch_norm normalises with
        invstd = torch.rsqrt(var + eps)
        ctx.group, ctx.hipdev = group, hipdev
        if hipdev:
            y = hip.batchnorm_apply(x, mean.contiguous(), invstd, weight, bias)
            ctx.save_for_backward(x, mean.contiguous(), invstd, weight, n)
        else:
            xhat = (x - mean) * invstd
            y = xhat * weight + bias
            ctx.save_for_backward(xhat, None, invstd, weight, n)
        ctx.mark_non_differentiable(mean, var, n)
        return y, mean, var, n

    @staticmethod
    def backward(ctx, dy, *_unused):
        x_or_xhat, mean, invstd, weight, n = ctx.saved_tensors
        C = dy.shape[1]
        if ctx.hipdev:
            from . import hip
            dy = Fh._unit_inner(dy)
            dw, db = hip.batchnorm_bwd_sums(x_or_xhat, dy, mean, invstd)
        else:
            dw, db = (dy * x_or_xhat).sum(0), dy.sum(0)
        s = torch.cat([db, dw])
        all_reduce(s, group=ctx.group)
        if ctx.hipdev:
            dx = hip.batchnorm_bwd_apply(x_or_xhat, dy, mean, invstd, weight, (s[:C] / n).contiguous(),
                                         (s[C:] / n).contiguous())
        else:
            dx = (weight * invstd) * (dy - s[:C] / n - x_or_xhat * (s[C:] / n))
        return dx, dw, db, None, None


class SyncBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d whose training-mode statistics span every rank's local batch, so that a model with BatchNorm
    towers (MMOE & co.) trained on G ranks equals the single-process run on the global batch.  Same
    parameters, buffers and state_dict keys as nn.BatchNorm1d; eval mode is the stock layer.  Device-agnostic torch
    ops on the CPU (gloo), the rp_batchnorm_* kernels + RCCL on HIP tensors."""

    group = None

    @classmethod
    def from_bn(cls, bn: nn.BatchNorm1d, group=None):
        m = cls(bn.num_features, eps=bn.eps, momentum=bn.momentum, affine=bn.affine,
                track_running_stats=bn.track_running_stats)
        m.load_state_dict(bn.state_dict())
        if bn.affine:
            m.weight, m.bias = bn.weight, bn.bias  # the very same Parameters (optimizers built earlier keep working)
        sd = bn.state_dict()
        if len(sd):
            m.to(next(iter(sd.values())).device)
        m.train(bn.training)
        m.group = group
        return m

    def forward(self, x):
        if not (self.training or not self.track_running_stats or self.running_mean is None) or x.dim() != 2:
            return super().forward(x)
        w = self.weight if self.affine else torch.ones(self.num_features, dtype=x.dtype, device=x.device)
        b = self.bias if self.affine else torch.zeros(self.num_features, dtype=x.dtype, device=x.device)
        y, mean, var, n = _SyncBatchNormFn.apply(x, w, b, self.eps, self.group)
        if self.training and self.track_running_stats and self.running_mean is not None:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                self.running_var.mul_(1 - mom).add_(var * (n / (n - 1).clamp_min(1.0)), alpha=mom)
        return y


def sync_batchnorm(model: nn.Module, group=None) -> nn.Module:
    """Replace every nn.BatchNorm1d of `model` by SyncBatchNorm1d (in place; same names, parameters and buffers)."""
    def swap(mod):
        for name, child in list(mod.named_children()):
            if type(child) is nn.BatchNorm1d:
                mod._modules[name] = SyncBatchNorm1d.from_bn(child, group)  # (also inside a ModuleList with named entries)
            else:
                swap(child)
    swap(model)
    return model


def shard_model_tables(model: nn.Module, world: int, rank: int, group=None) -> nn.Module:
    """Replace every EmbeddingLayer of `model` by its row-sharded counterpart (in place).  Build the model
    identically on every rank first (same seed): the shards are cut from that common initialisation."""
    from .models.layers.embedding import EmbeddingLayer

    def swap(mod):
        for name, child in list(mod.named_children()):
            if isinstance(child, EmbeddingLayer):
                setattr(mod, name, ShardedEmbeddingLayer(child, world, rank, group))
            else:
                swap(child)
    swap(model)
    if world > 1:
        sync_batchnorm(model, group)  # BatchNorm towers: statistics over the global batch
    return model


class sharded_construction:
    """Context manager: models constructed inside it get ShardedEmbeddingLayers built shard-locally
    (`ShardedEmbeddingLayer.from_spec`) wherever the single-process model has an EmbeddingLayer."""

    def __init__(self, world: int, rank: int, group=None, device=None):
        self.spec = dict(world=world, rank=rank, group=group, device=device)

    def __enter__(self):
        from .models.layers import embedding as E
        self._prev = E._SHARD_SPEC
        E._SHARD_SPEC = self.spec
        return self

    def __exit__(self, *exc):
        from .models.layers import embedding as E
        E._SHARD_SPEC = self._prev
        return False


def build_sharded_model(factory, world: int, rank: int, device=None, seed=None, group=None) -> nn.Module:
    """`factory()` builds the model (e.g. lambda: DeepFM(...)); every rank calls this with the same seed.  The tables
    are created row-sharded from the start — per-rank memory is arena / world plus one temporary table, never the full
    arena — with the same initial values the single-process construction draws; dense parameters are replicated."""
    if seed is not None:
        torch.manual_seed(seed)
    with sharded_construction(world, rank, group, device):
        if device is not None:
            with torch.device(device):
                model = factory()
        else:
            model = factory()
    if world > 1:
        sync_batchnorm(model, group)
    return model


def dense_parameters(model: nn.Module):
    sharded = {id(m.local_arena) for m in model.modules() if isinstance(m, ShardedEmbeddingLayer)}
    return [p for p in model.parameters() if id(p) not in sharded]


def allreduce_dense_grads(model: nn.Module, group=None):
    """Average the replicated parameters' gradients over the ranks in one flat bucket."""
    ps = [p for p in dense_parameters(model) if p.grad is not None]
    if ps:
        lib_copy = False
        if ps[0].grad.is_cuda and all(p.grad.is_contiguous() and p.grad.dtype is torch.float32 for p in ps):
            # the bucket is a persistent buffer filled and emptied by ONE library launch each (rp_multi_copy): torch.cat and
            # the per-tensor copy_ back are ATen launches, which a recorded step (launch plan) must not hold
            from . import hip
            total = sum(p.numel() for p in ps)
            flat = getattr(model, "_rp_flat_grads", None)
            if flat is None or flat.numel() != total or flat.device != ps[0].grad.device:
                flat = torch.empty((total,), dtype=torch.float32, device=ps[0].grad.device)
                object.__setattr__(model, "_rp_flat_grads", flat)
            views, off = [], 0
            for p in ps:
                views.append(flat[off:off + p.numel()].view_as(p.grad))
                off += p.numel()
            lib_copy = hip.multi_copy(views, [p.grad for p in ps])
        if not lib_copy:
            flat = torch.cat([p.grad.reshape(-1) for p in ps])
        all_reduce(flat, group=group)
        if not _SEED_SCALED[0] and dist.get_world_size(group) > 1:  # (a recorded step's backward is seeded with 1 / world: the sum IS the mean)
            flat /= dist.get_world_size(group)
        if lib_copy:
            hip.multi_copy([p.grad for p in ps], views)
        else:
            off = 0
            for p in ps:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
    # this step's collectives are all enqueued: the announced next batch may now request its rows (side stream)
    for m in model.modules():
        if isinstance(m, ShardedEmbeddingLayer):
            m._request_ahead()
