"""EmbeddingLayer — drop-in for rec_pangu/models/layers/embedding.py:11-71, MI355X-first.

Same constructor, same `emb_feature` order (enc_dict key order of entries with 'vocab_size'),
same ModuleDict of nn.Embedding(vocab_size+1, D) and therefore the same state_dict keys
(`embedding_layer.<col>.weight`) and the same RNG draws at construction.

What is different is where the rows live: all tables of the layer are views into ONE contiguous
arena [sum(vocab+1), D] (HBM layout for the fused gather: a lookup is arena[row_base[f]+id]), and
their gradients are views into one dense gradient arena that the HIP backward writes in place.
`module.to()/cuda()/float()` move the arena once and re-point the per-table Parameters; anything
that replaces a table tensor behind our back (set_weights, external .data assignment) is detected
at the next forward and the arena is re-packed.

Device dispatch is by where the arena lives: a HIP-resident layer ALWAYS runs the HIP kernels
(rec_pangu_amd/hip.py raises if the library is missing — no fallback); a CPU-resident layer is
BASELINE.json config 0 ("plumbing, no GPU") and uses plain torch ops.
"""
import copy
import os
import weakref
from typing import Dict, List, Optional, Union

import torch
from torch import nn

from ... import functional as Fh

# the last two (id tensors, versions, arena signature) -> (keys, sorted keys, positions)[, side-stream event]: the batch in
# flight and the one whose sort was started ahead (EmbeddingLayer._sorted_keys / prefetch_sort)
_SORT_CACHE: list = []
_SIDE_STREAMS: dict = {}  # device -> the stream sorts started ahead run on
# PINNED entries (graph_step.GraphedTrainStep): the id tensors are STATIC input buffers that are refilled in place, the
# (keys, sorted keys, positions) tensors are persistent and re-sorted IN PLACE by prefetch_sort — a captured hipGraph
# holds all their addresses.  Matched by identity only (a refill bumps the versions), never evicted, no events (the
# order is the capture's: the side stream is joined before the capture ends).
_SORT_PINNED: list = []
# id(keys tensor of a pinned entry) -> its persistent sort workspace: a captured step's re-sort must not allocate (its
# launches may be re-issued on a side stream, where memory of the capture's pool could alias the step's temporaries)
_PINNED_WS: dict = {}
_PINNED_MARKS: dict = {}  # id(pinned key buffer) -> the duplicate marks / unique-row lists are made with its sort

# id(sorted-keys tensor) -> [weakref(sorted keys), (B, fields), (dupq, dupkeys)]: the marks rp_embed_grad_smp wants, made
# right behind the sort that made the keys (EmbeddingLayer._mark_sorted: on the sort's stream, one step ahead when the sort
# is) and re-made in place when a pinned entry is re-sorted in place
_SMP_MARKS: dict = {}



class DeferredGradView(torch.Tensor):
    """`.grad` of an embedding table while FusedAdam(defer=True) drives it (the contract was prose).

    Under the deferred execution a row's real optimizer step waits in the gradient arena until the row is next looked up, so
    between backward() and step() a table's gradient rows hold THIS step's gradients AND gradients of earlier steps that are
    still waiting — not the reference's dense gradient (rec_pangu/trainer.py:60-76 never reads it).  Anything that computes
    with such a tensor (gradient clipping, a norm, an in-place rescale) would silently read or damage waiting gradients:
    every torch operation on it RAISES instead.  Addresses, shapes and dtypes stay readable (the library checks that a
    table's .grad is still its arena view that way); `with rec_pangu_amd.models.layers.embedding.deferred_grad_reads():`
    opens the view for code that knows what the rows hold (tests, diagnostics); make_adam(..., defer=False) /
    RP_ADAM_DEFER=0 gives plain dense gradients."""
    _open = 0
    _META = {"data_ptr", "size", "stride", "dim", "numel", "nelement", "element_size", "is_contiguous", "storage_offset",
             "__get__", "__repr__", "__str__", "__format__", "__len__", "__hash__", "__reduce_ex__", "__deepcopy__",
             "is_floating_point", "is_complex", "get_device", "untyped_storage", "type", "requires_grad_", "_is_view"}

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        if cls._open > 0 or name in cls._META:
            with torch._C.DisableTorchFunctionSubclass():
                out = func(*args, **(kwargs or {}))
            return out
        raise RuntimeError(
            f"torch.{name} on the .grad of an embedding table driven by FusedAdam(defer=True): its rows hold this step's "
            "gradients AND gradients of earlier steps whose optimizer step is still waiting (rec_pangu_amd/optim.py, deferred "
            "execution) — not a dense gradient.  Use make_adam(model, lr, defer=False) (or RP_ADAM_DEFER=0) when table gradients "
            "are read or rewritten between backward() and step() (gradient clipping), or wrap a knowing read in "
            "rec_pangu_amd.models.layers.embedding.deferred_grad_reads()")


class deferred_grad_reads:
    """context manager: table .grad views behave as plain tensors inside (see DeferredGradView)"""

    def __enter__(self):
        DeferredGradView._open += 1
        return self

    def __exit__(self, *exc):
        DeferredGradView._open -= 1
        return False


# set by rec_pangu_amd.sharded.sharded_construction(): models built inside it get row-sharded embedding layers that
# only ever allocate their own shard (see make_embedding_layer)
_SHARD_SPEC = None


def make_embedding_layer(enc_dict, embedding_dim: int):
    """What BaseModel / LR_Layer call to create their embedding layer: the arena-backed EmbeddingLayer, or — inside
    `sharded.sharded_construction(world, rank)` — a ShardedEmbeddingLayer holding rows r % world == rank, initialised
    with the very same RNG draws (table by table through a temporary) without ever allocating the full arena."""
    if _SHARD_SPEC is None:
        return EmbeddingLayer(enc_dict=enc_dict, embedding_dim=embedding_dim)
    from ...sharded import ShardedEmbeddingLayer
    return ShardedEmbeddingLayer.from_spec(enc_dict, embedding_dim, **_SHARD_SPEC)


class EmbeddingLayer(nn.Module):
    def __init__(self, enc_dict: Dict[str, Dict[str, Union[int, str]]], embedding_dim: int) -> None:
        super().__init__()
        self.enc_dict = enc_dict
        self.embedding_dim = embedding_dim
        self.embedding_layer = nn.ModuleDict()
        self.emb_feature: List[str] = []
        # "sync": raise IndexError right after the gather (one 4-byte D2H per forward, what the
        # reference's CPU nn.Embedding does); "deferred": only when raise_if_bad_index() is called.
        self.check_indices = "sync"

        rows = [enc_dict[c]["vocab_size"] + 1 for c in enc_dict.keys() if "vocab_size" in enc_dict[c].keys()]
        arena = torch.empty((sum(rows), embedding_dim))
        off = 0
        for col in self.enc_dict.keys():
            if "vocab_size" in self.enc_dict[col].keys():
                r = self.enc_dict[col]["vocab_size"] + 1
                self.emb_feature.append(col)
                emb = nn.Embedding(r, embedding_dim, _weight=arena[off:off + r])
                with torch.no_grad():
                    nn.init.normal_(emb.weight)  # the draw nn.Embedding.reset_parameters() would make
                self.embedding_layer.update({col: emb})
                off += r
        self._arena = arena
        self._grad_arena: Optional[torch.Tensor] = None
        self._touched: Optional[torch.Tensor] = None  # sorted keys written by the last backward
        self._grad_clean = True  # gradient arena known to be all zero
        self._noclear_ok = True  # every gradient launch of this layer writes ALL rows of its key list (LazyAdamRows.replay)
        self._dev_meta = None
        self._lazy = None        # optim.LazyAdamRows when the optimiser runs the exact lazy dense Adam
        self._presorted = None   # (keys, sorted keys, sorted positions) of the batch being looked up
        self._fm_link = None     # functional.FMFold of the last gather that produced an FM term
        self._tag_tables()
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_lazy())
        # loading weights into a layer whose optimizer runs lazily: bring every row to the current step first, so that
        # no skipped zero-gradient step is replayed on top of the loaded values later
        self.register_load_state_dict_pre_hook(lambda module, *a, **k: module.flush_lazy())

    # ------------------------------------------------------------------ arena bookkeeping
    def _tables(self):
        """The per-table Parameters in field order.  Cached: walking the ModuleDict costs ~100 us per call at 26
        tables and this runs several times per step; the cache is dropped when a table Parameter object is replaced
        (set_weights, load with assign=True, ...), which the identity check below detects."""
        mods = self.__dict__.get("_tab_mods")
        tabs = self.__dict__.get("_tab_cache")
        if mods is not None:
            for m, p in zip(mods, tabs):
                if m._parameters["weight"] is not p:
                    mods = None
                    break
        if mods is None:
            mods = [self.embedding_layer[c] for c in self.emb_feature]
            tabs = [m.weight for m in mods]
            self.__dict__["_tab_mods"], self.__dict__["_tab_cache"] = mods, tabs
        return tabs

    def table_parameters(self):
        return self._tables()

    def _tag_tables(self):
        ref = weakref.ref(self)
        for p in self._tables():
            p._rp_store = ref  # lets FusedAdam find the arena behind a table Parameter

    def _point_at(self, arena: torch.Tensor):
        off = 0
        for p in self._tables():
            r = p.shape[0]
            p.data = arena[off:off + r]
            off += r
        self._arena = arena
        self._dev_meta = None
        self._tag_tables()

    def __deepcopy__(self, memo):
        """copy.deepcopy(model): nn.Parameter.__deepcopy__ clones every table into a storage of its own, which would
        tear the tables off the arena — and, with the lazy optimizer active, make the next forward re-pack the arena from
        table values that still owe their skipped steps while the moment arenas are stamped current.  The clone gets
        ONE copy of the arena (and of the gradient / moment arenas) and its tables are pointed back into it."""
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__({k: copy.deepcopy(v, memo) for k, v in self.__dict__.items()})
        new._point_at(new._arena)  # (same values: the arena copy and the table clones were taken from the same memory)
        if new._grad_arena is not None and self._grads_are_ours():
            new._attach_grads()
        return new

    def _apply(self, fn, recurse=True):
        # one move for the whole arena instead of one per table (module.to / .cuda / .float ...)
        self._ensure_packed()
        self.flush_lazy()
        self._point_at(fn(self._arena))
        if self._lazy is not None:
            self._lazy.apply(fn)
        if self._grad_arena is not None:
            self._grad_arena = fn(self._grad_arena)
            self._attach_grads()
        self._touched = None if self._touched is None else fn(self._touched)
        if self.__dict__.get("_shadow") is not None:
            self._shadow_stamp = None  # (rebuilt from the moved tables at the next training forward)
        return self

    def _ensure_packed(self):
        a, D = self._arena, self.embedding_dim
        off, ok = 0, True
        for p in self._tables():
            r = p.shape[0]
            if (p.device != a.device or p.dtype != a.dtype or p.dim() != 2 or p.shape[1] != D
                    or not p.is_contiguous() or off + r > a.shape[0]
                    or p.data_ptr() != a.data_ptr() + off * D * a.element_size()):
                ok = False
                break
            off += r
        if ok and off == a.shape[0]:
            tabs = self._tables()
            if tabs and (getattr(tabs[0], "_rp_store", None) is None or tabs[0]._rp_store() is not self):
                self._tag_tables()  # copy.deepcopy(model): the clone's tables still carried the original's weakref
            return
        # a table was replaced behind our back (set_weights, .data assignment, load with assign=True): re-pack.
        # Rows the lazy optimizer still owes steps to are brought up to date in the OLD arena first (the replaced table
        # is not in it any more, the others are); the moment arenas survive a same-shape re-pack with every row
        # stamped current, and are dropped (rebuilt by the optimizer) when the shape changed.
        if self._lazy is not None:
            self._lazy.flush(self)
        with torch.no_grad():
            tabs = self._tables()
            arena = torch.cat([p.detach().to(device=a.device, dtype=a.dtype).reshape(p.shape[0], -1) for p in tabs],
                              dim=0).contiguous()
        self.embedding_dim = arena.shape[1]
        self._point_at(arena)
        self._grad_arena, self._touched, self._grad_clean = None, None, True
        self._rows_sig_cache = None
        if self._lazy is not None and self._lazy.m.shape != arena.shape:
            self._lazy = None

    @property
    def arena(self) -> torch.Tensor:
        return self._arena

    def _meta(self):
        dev = self._arena.device
        if self._dev_meta is None or self._dev_meta[0].device != dev:
            rows = [p.shape[0] for p in self._tables()]
            base = [0]
            for r in rows[:-1]:
                base.append(base[-1] + r)
            self._dev_meta = (torch.tensor(base, dtype=torch.int64, device=dev),
                              torch.tensor(rows, dtype=torch.int64, device=dev),
                              torch.zeros((1,), dtype=torch.int32, device=dev),
                              max(1, int(sum(rows) - 1).bit_length()))
        return self._dev_meta

    row_base = property(lambda self: self._meta()[0])
    row_count = property(lambda self: self._meta()[1])
    err_flag = property(lambda self: self._meta()[2])

    def raise_if_bad_index(self):
        """Turn the device-side out-of-range flag into the reference's IndexError (synchronises)."""
        if self._dev_meta is not None and self._arena.is_cuda and int(self._dev_meta[2].item()) != 0:
            self._dev_meta[2].zero_()
            raise IndexError("index out of range in self")

    # ------------------------------------------------------------------ gradients (HIP path)
    def _attach_grads(self):
        off = 0
        guard = self._lazy is not None and bool(getattr(self._lazy, "defer", False)) and os.environ.get("RP_GRAD_GUARD", "1") != "0"
        for p in self._tables():
            r = p.shape[0]
            if p.requires_grad:  # a frozen table (set_weights(trainable=False)) never shows a gradient
                view = self._grad_arena[off:off + r]
                # (deferred lazy Adam: the rows of other steps wait in this arena — computing with the view raises)
                p.grad = view.as_subclass(DeferredGradView) if guard else view
            off += r

    @property
    def grad_arena(self):
        return self._grad_arena

    def grads_are_arena(self) -> bool:
        """True when every table's .grad is the matching view of the gradient arena."""
        return self._grads_are_ours() and all(p.requires_grad for p in self._tables())

    def grads_were_zeroed(self):
        """FusedAdam(fuse_zero_grad) cleared the whole gradient arena."""
        self._touched, self._grad_clean = None, True

    def _grads_are_ours(self) -> bool:
        g, D = self._grad_arena, self.embedding_dim
        if g is None:
            return False
        off = 0
        for p in self._tables():
            if p.requires_grad and (p.grad is None or p.grad.data_ptr() != g.data_ptr() + off * D * 4
                                    or p.grad.shape != p.shape):
                return False
            off += p.shape[0]
        return True

    def flush_lazy(self):
        """Bring every row to the optimiser's current step (no-op unless the lazy Adam is active).  Called before
        anything reads the raw tables: state_dict(), .to(), checkpoints."""
        if self._lazy is not None:
            self._lazy.flush(self)

    def _tiny_tables(self):
        """[(field, first arena row, rows), ...] of the tables rp_embed_grad_tiny takes (D = 64; each <= 254 rows, the
        smallest first while they fit 224 accumulator rows, at most 16), or None; RP_GRAD_TINY=0 turns the path off"""
        hit = self.__dict__.get("_tiny_cache")
        sig = self._rows_sig()
        if hit is not None and hit[0] is sig:
            return hit[1]
        out = None
        if self.embedding_dim == 64 and os.environ.get("RP_GRAD_TINY", "1") != "0" and len(sig) <= 64:
            order = sorted(range(len(sig)), key=lambda f: sig[f])
            pick, total = [], 0
            for f in order:
                if sig[f] <= 254 and total + sig[f] <= 224 and len(pick) < 16:
                    pick.append(f)
                    total += sig[f]
            if total >= 1 and len(pick) >= 2:  # (a single tiny table is not worth two extra launches)
                out = [(f, sum(sig[:f]), sig[f]) for f in sorted(pick)]
        self.__dict__["_tiny_cache"] = (sig, out)
        return out

    def _smp_tables(self, B: int):
        """[(field, first arena row, rows), ...] of the BIG tables rp_embed_grad_smp takes (round 6: the sample-major form of
        the first layer's backward — tables of at least RP_SMP_MIN x B rows, where most runs of the sorted pair list are
        single pairs; not rp_embed_grad_tiny's; < 2^24 rows each; the 16 largest), or None; RP_GRAD_SMP=0 turns the path off"""
        hit = self.__dict__.get("_smp_cache")
        sig = self._rows_sig()
        if hit is not None and hit[0] is sig and hit[1] == B:
            return hit[2]
        out = None
        # (batches below RP_SMP_MIN_BATCH keep round 5's single row-sorted launch: the three-form backward is 16 launches more,
        #  and a b = 8192 step — the per-GPU batch of a strong-scaling run — is bound by launches: 0.50 ms against 0.43)
        if self.embedding_dim == 64 and os.environ.get("RP_GRAD_SMP", "1") != "0" and len(sig) <= 64 \
                and B >= int(os.environ.get("RP_SMP_MIN_BATCH", "32768")):
            tiny = {t[0] for t in (self._tiny_tables() or ())}
            need = float(os.environ.get("RP_SMP_MIN", "1.0")) * B
            pick = sorted((f for f in range(len(sig)) if f not in tiny and need <= sig[f] < (1 << 24)),
                          key=lambda f: -sig[f])[:16]
            if pick and len(pick) * B < (1 << 24) and sum(sig) < (1 << 31):
                out = [(f, sum(sig[:f]), sig[f]) for f in sorted(pick)]
        self.__dict__["_smp_cache"] = (sig, B, out)
        return out

    def _mark_sorted(self, sk, sp, force: bool = False) -> None:
        """behind a sort of this layer's pairs (on the stream that sorted them): the duplicate marks of the big tables.  Only for
        a layer whose backward has taken the three-form path before (`_marks_wanted`, set by accumulate_grad) — DCN, xDeepFM,
        AutoInt and MMOE reduce through rp_embed_grad_reduce and made six launches' worth of marks per step for nobody."""
        from ... import hip
        if not (force or getattr(self, "_marks_wanted", False)):
            return
        F = len(self.emb_feature)
        if sk.numel() == 0 or sk.numel() % F:
            return
        B = sk.numel() // F
        smp = self._smp_tables(B)
        if smp is None:
            return
        ent = _SMP_MARKS.get(id(sk))
        sig = (B, tuple(smp))
        same = ent is not None and ent[0]() is sk and ent[1] == sig
        out = ent[2] if same else None
        marks = hip.embed_grad_smp_mark(sk, sp, B, [t[0] for t in smp], out=out)
        # ... and the unique-row lists of the mid-size tables (every field that is neither tiny nor big: rp_embed_grad_ss)
        skip = 0
        for f, _, _ in list(self._tiny_tables() or ()) + list(smp):
            skip |= 1 << f
        ss = None
        if skip != (1 << F) - 1 and os.environ.get("RP_GRAD_SS", "1") != "0" and os.environ.get("RP_SS_MARK_AHEAD", "1") != "0":
            ss_out = ent[3][1] if (same and len(ent) > 3 and ent[3] is not None and ent[3][0] == skip) else None
            ss = (skip, hip.embed_grad_ss_mark(sk, B, skip, out=ss_out))
        if out is None:
            if len(_SMP_MARKS) >= 8:
                for k in [k for k, e in _SMP_MARKS.items() if e[0]() is None] or list(_SMP_MARKS)[:4]:
                    _SMP_MARKS.pop(k, None)
            _SMP_MARKS[id(sk)] = [weakref.ref(sk), sig, marks, ss]
        else:
            ent[3:] = [ss]

    def _marks_of(self, sk, sp, smp, B: int):
        ent = _SMP_MARKS.get(id(sk))
        self._marks_wanted = True
        if ent is None or ent[0]() is not sk or ent[1] != (B, tuple(smp)):
            self._mark_sorted(sk, sp, force=True)  # (a sort nobody marked: an eager backward that sorted for itself)
            ent = _SMP_MARKS[id(sk)]
        return ent[2]

    def _ss_marks_of(self, sk, skip: int):
        """the unique-row lists made with the sort for exactly these kept fields, or None (rp_embed_grad_ss makes its own)"""
        ent = _SMP_MARKS.get(id(sk))
        if ent is None or ent[0]() is not sk or len(ent) < 4 or ent[3] is None or ent[3][0] != skip:
            return None
        return ent[3][1]

    def accumulate_grad(self, keys, B: int, dx, gfm, ssum, presorted=None, fused=None, pool=None, plan_keep=None, seg=None,
                        seg_first: bool = False, fork2=None):
        """Called from the autograd node of the gather: dense table gradients, reference semantics
        (aten::embedding_dense_backward: every table gets a full [V+1, D] gradient, zeros where no
        sample looked).  Invariant kept between steps: the gradient arena is zero everywhere except
        the rows listed in `_touched`, so a fresh gradient costs a sparse re-zero, not an 8.6 GB fill."""
        from ... import hip
        D = self.embedding_dim
        fresh = not self._grads_are_ours()
        if self._grad_arena is None or self._grad_arena.shape != self._arena.shape \
                or self._grad_arena.device != self._arena.device:
            self._grad_arena = torch.zeros_like(self._arena)
            self._touched, self._grad_clean = None, True
        elif fresh and not self._grad_clean:
            # zero_grad() dropped the .grad views: clear the rows the previous backward wrote
            if self._touched is not None:
                hip.zero_rows(self._touched, D, self._grad_arena)
            else:
                self._grad_arena.zero_()
            self._touched, self._grad_clean = None, True
        if pool is not None and presorted is None:  # pooled multi-id lookup of one table: keys of its flat id list
            f = pool[4]
            keys = hip.embed_keys(self.row_base[f:f + 1], self.row_count[f:f + 1], [pool[5]], self.err_flag)
        if presorted is not None:
            sk, sp = presorted
        else:
            sk, sp = hip.sort_pairs(keys, end_bit=self._meta()[3])
        if self._lazy is not None and self._lazy._noclear is not None:
            # the catch-up launch in front of this step's forward left its applied gradient rows uncleared, counting on
            # THIS launch to overwrite them (LazyAdamRows.replay): kept if it writes that key list without accumulating
            self._lazy.resolve_noclear(self, sk if self._grad_clean else None)
        if pool is not None:  # (g [B, D], 1 / count or None, bag of every id or None, ids per dense bag, field, ids)
            hip.embed_pool_bwd(sk, sp, D, pool[0], pool[1], pool[2], pool[3], self._grad_arena,
                               accumulate=not self._grad_clean)
        elif fused is not None:  # (dH, W^T) of the Linear that consumes x: its dgrad is formed inside the reduce
            skip = 0
            smp = None
            tiny = self._tiny_tables() if (keys is not None and dx is None and keys.numel() == len(self.emb_feature) * B) else None
            for f, _, _ in (tiny or ()):
                skip |= 1 << f

            def run_tiny():
                # the tables of a few rows (Criteo: 8 fields, 31 % of the pairs) take the sample-major one-hot path; the
                # row-sorted kernel leaves their fields out.  Inside a recorded launch plan these launches join the first
                # layer's side work on the plan's second side stream (functional._EmbedGatherLinear.backward)
                in_plan = plan_keep is not None
                if in_plan:
                    hip.LaunchPlan.section(2)
                try:
                    hip.embed_grad_tiny(keys, B, tiny, fused[0], fused[1], gfm, ssum, self._arena, self._grad_arena,
                                        accumulate=not self._grad_clean, keep=plan_keep, dw=None if seg is None else seg[1])
                finally:
                    if in_plan:
                        hip.LaunchPlan.section(0)

            if tiny and not seg_first:
                run_tiny()
            if seg is not None:
                # (weight [64, K], dw [64, K]) of the consuming Linear: segment sums first, one matrix pass per run that also
                # yields the embedding columns of dw — the forward stored no activation (functional._EmbedGatherLinear)
                if keys is None or dx is not None or keys.numel() != len(self.emb_feature) * B:
                    raise RuntimeError("the fused first layer stored no activation, but its backward is not the field-major "
                                       "single-device form rp_embed_grad_seg covers")
                # round 6: the BIG tables (runs of the sorted list are mostly single pairs: the sort buys nothing and the
                # (field, row) order re-gathers every dH / S row once per field) go through the batch in sample order
                # instead (rp_embed_grad_smp); its main launch first, then — forked from that point inside a recorded plan —
                # rp_embed_grad_seg over the remaining fields with the launches behind the sample-major one (the weight
                # gradient's partial sums, the duplicate runs: other rows than anything beside them) on the second stream
                smp = self._smp_tables(B) if hip.embed_grad_smp_fits(D, 64, fused[0]) else None
                in_plan = plan_keep is not None
                acc = not self._grad_clean
                if smp:
                    for f, _, _ in smp:
                        skip |= 1 << f
                rest = skip != (1 << len(self.emb_feature)) - 1   # fields left for the row-sorted form
                # ... and the row-sorted form for the remaining (mid-size) tables in two launches (rp_embed_grad_ss: a streaming
                # segment-sum launch, then the matrix launch over the unique rows).  (The segment-sum launch uses no LDS, but it
                # cannot run BESIDE the sample-major launch on a second stream: 4 x 112 and 2 x 240 registers per SIMD lane do
                # not fit the file of 512 together — one after the other on the main stream.)
                ss = bool(smp) and rest and os.environ.get("RP_GRAD_SS", "1") != "0"
                seg_args = (sk, sp, B, D, fused[0], seg[0], gfm, ssum, self._arena, self._grad_arena)
                seg_kw = dict(accumulate=acc, skip_fields=skip, field_rows=self._rows_sig(), dw=seg[1], keep=plan_keep)
                smp_ws = None
                # RP_TINY_EARLY=1 (a recorded plan only): the tiny tables' launches are forked IN FRONT of the sample-major
                # launch and run beside it on the second stream; the launches behind the sample-major one follow them there
                # (rp_plan_side2_sync: that stream waits for the sample-major launch at that point)
                tiny_early = bool(smp and tiny and seg_first and in_plan and fork2 is not None
                                  and os.environ.get("RP_TINY_EARLY", "1") == "1")
                if tiny_early:
                    fork2()
                if smp:
                    marks = self._marks_of(sk, sp, smp, B)
                    smp_args = (keys, marks, B, len(self.emb_feature), smp, fused[0], seg[0], gfm, ssum, self._arena,
                                self._grad_arena)
                    smp_ws = hip.embed_grad_smp(*smp_args, accumulate=acc, dw=seg[1], keep=plan_keep, phases=1)
                if ss:  # (the unique-row lists made with the sort — or just now, by _marks_of, for a sort nobody had marked)
                    seg_kw["marks"] = self._ss_marks_of(sk, skip)
                if tiny_early:
                    run_tiny()
                    hip.LaunchPlan.side2_sync()
                    tiny = None
                elif fork2 is not None:
                    fork2()
                # (inside a recorded plan the workspaces stay referenced until the join: the side launches issued BEHIND these
                #  run beside them on another stream, and the capture's one-stream allocator would hand them their memory)
                if ss:
                    hip.embed_grad_ss(*seg_args, **seg_kw)
                elif rest:
                    hip.embed_grad_seg(*seg_args, **seg_kw)
                if smp:
                    # the launches behind the sample-major one: other rows than anything beside them — on the second stream,
                    # forked at the mark above (behind the sample-major launch), beside the launches of the row-sorted form; in
                    # ISSUE order behind those: issued in front of them, the 4096 short workgroups of the duplicate reduce held
                    # the segment-sum launch up by 56 us (profiles/r06 trace notes)
                    # (RP_SMP_BEHIND=main: on the main stream behind everything else of this phase — with the tiny tables'
                    #  launches in front of them the second stream is the longer one, the main stream idles at the join)
                    side = in_plan and os.environ.get("RP_SMP_BEHIND", "side") != "main"
                    if side:
                        hip.LaunchPlan.section(2)
                    try:
                        hip.embed_grad_smp(*smp_args, accumulate=acc, dw=seg[1], phases=2, ws=smp_ws)
                    finally:
                        if side:
                            hip.LaunchPlan.section(0)
            else:
                if fork2 is not None:
                    fork2()
                hip.embed_grad_gemm(sk, sp, B, D, fused[0], fused[1], dx, gfm, ssum, self._arena, self._grad_arena,
                                    accumulate=not self._grad_clean, skip_fields=skip)
            if tiny and seg_first:
                # (behind the long main-stream launch in issue order: see rp_plan_fork2_mark.  Round 6, with the sample-major
                #  launch in front: on the MAIN stream — the second stream's launches cannot run beside rp_embed_grad_smp /
                #  _seg, whose workgroups hold all of every CU's LDS, so whatever sits there runs in the step's tail one short
                #  launch after another while the main stream idles at the join: the tail is split over both streams)
                if smp and os.environ.get("RP_TINY_MAIN", "1") == "1":
                    hip.embed_grad_tiny(keys, B, tiny, fused[0], fused[1], gfm, ssum, self._arena, self._grad_arena,
                                        accumulate=not self._grad_clean, keep=plan_keep, dw=None if seg is None else seg[1])
                else:
                    run_tiny()
        else:
            hip.embed_grad_reduce(sk, sp, B, D, dx, gfm, ssum, self._arena, self._grad_arena,
                                  accumulate=not self._grad_clean)
        if self._touched is None:
            self._touched, self._touched_unsorted = sk, False
        else:  # several backward passes before one optimiser step: the union is no longer sorted
            self._touched, self._touched_unsorted = torch.cat([self._touched, sk]), True
        self._grad_clean = False
        if fresh:
            self._attach_grads()

    # ------------------------------------------------------------------ reference API
    def set_weights(self, col_name: str, embedding_matrix: torch.Tensor, trainable: Optional[bool] = True) -> None:
        """embedding.py:36-47 — replace one table (re-packed into the arena at the next forward)."""
        self.embedding_layer[col_name].weight = nn.Parameter(embedding_matrix)
        if not trainable:
            self.embedding_layer[col_name].weight.requires_grad = False

    def _idx_list(self, X):
        out = []
        for c in self.emb_feature:
            t = X[c]
            if t.dtype is not torch.int64 or t.dim() != 1 or not t.is_contiguous():
                t = t.long().reshape(-1).contiguous()
            out.append(t)
        return out

    def gather_concat(self, X, dense: List[torch.Tensor], want_fm: bool, pad_to: int = 64):
        """HIP path: (x [B, ldx], fm [B,1] or None).  x = embeddings (F*D) | dense (ND) | zero pad."""
        self._ensure_packed()
        return self._gather(self._idx_list(X), None, dense, want_fm, pad_to, src=tuple(X[c] for c in self.emb_feature))

    def gather_linear_fits(self, n_dense: int, linear: nn.Linear, pad_to: int = 64) -> bool:
        """can lookup + concat + FM + `linear` (+ ReLU) run as the one fused launch (rp_embed_gather_linear_fwd)?"""
        from ... import hip
        F, D = len(self.emb_feature), self.embedding_dim
        d = F * D + n_dense
        ldx = (d + pad_to - 1) // pad_to * pad_to
        w = linear.weight
        # (the answer only depends on shapes, alignment and the matrix-core mode: remembered per weight tensor — the check
        #  runs in every forward and used to stage the weight and cross the C ABI each time)
        key = (id(w), w.data_ptr() % 16, tuple(w.shape), F, D, n_dense, pad_to, self._arena.is_cuda, hip.get_matmul_precision())
        hit = self.__dict__.get("_glf_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        ok = (os.environ.get("RP_GATHER_LINEAR", "1") != "0" and self._arena.is_cuda and D == 64 and linear.out_features == 64 and linear.in_features == d
              and hip.get_matmul_precision() != "fp32"
              and hip.embed_gather_linear_fits(D, F, n_dense, 64, ldx, Fh._rows16(w)))
        self.__dict__["_glf_cache"] = (key, ok)
        return ok

    def bf16_lookup(self, enable: bool = True) -> None:
        """bf16-STORAGE inference (SURVEY D6's secondary mode): snapshot the tables into a bf16 copy of the arena (half the
        bytes: 4.3 GB at Criteo shape) that no-grad forwards of the fused lookup + first layer then read instead of the fp32
        arena — half the gather traffic, fp32 accumulation, logits within 6e-2 of the fp32 tables' (measured 3.7e-2; bf16 rounding of every
        looked-up value; outside the 1e-4 parity gate, which the fp32 tables keep).  The snapshot is taken of the CURRENT
        weights (owed optimizer steps flushed first); training on makes it stale — a forward that would read a stale
        snapshot raises.  enable=False drops it."""
        if not enable:
            self._arena_bf16 = None
            return
        self._ensure_packed()
        self.flush_lazy()
        self._arena_bf16 = self._arena.detach().to(torch.bfloat16)
        self._bf16_stamp = self._bf16_stamp_now()

    def bf16_training(self, enable: bool = True) -> None:
        """bf16-STORAGE TRAINING (SURVEY D6's perf mode with a stated tolerance; never the parity path): the fused lookup +
        first layer gathers from a bf16 LOOKUP COPY of the tables (half the row bytes) and stores its activation as bf16
        (half the bytes the weight gradient reads); the fp32 tables stay the master the optimizer works on, moments and
        every accumulation stay fp32.  The deferred table optimizer keeps the copy current: whenever rp_lazy_adam_catchup /
        _flush_deferred write a parameter row they write its round-to-nearest-even bf16 image too (needs
        make_adam(defer=True), the default; other optimizers on the tables raise).  Anything that changes the tables through
        torch (load_state_dict, set_weights, .to()) is noticed and the copy rebuilt.  Tolerance: the looked-up values are
        bf16-rounded (2^-9 relative): logits within 6e-2 of the fp32 tables' (tests/test_hip_models.py)."""
        if not enable:
            self._shadow = None
            return
        self._ensure_packed()
        if not self._arena.is_cuda or self.embedding_dim != 64:
            raise RuntimeError("bf16_training: a HIP-resident layer with embedding_dim 64 (the fused lookup + first layer)")
        self._shadow_rebuild()

    def _shadow_rebuild(self):
        self.flush_lazy()
        self._shadow = self._arena.detach().to(torch.bfloat16)
        self._shadow_stamp = self._shadow_stamp_now()

    def _shadow_stamp_now(self):
        # (torch-side writes bump these counters; the fused optimizer writes through raw pointers and keeps the copy itself)
        return (self._arena.data_ptr(), self._arena._version, tuple(p._version for p in self._tables()))

    def _shadow_for_training(self):
        sh = self.__dict__.get("_shadow")
        if sh is None:
            return None
        if sh.device != self._arena.device or sh.shape != self._arena.shape or self._shadow_stamp != self._shadow_stamp_now():
            self._shadow_rebuild()
            sh = self._shadow
        return sh

    def _bf16_stamp_now(self):
        """what the snapshot is valid for: torch's version counter of the arena, the lazy optimizer's step AND the
        library's weight epoch — the fused optimizers write the arena through raw pointers, which torch's counter does
        not see (FusedAdam(lazy_tables=False), torch.optim.Adam on the table views: ADVICE r3)"""
        from ... import hip
        return (self._arena._version, None if self._lazy is None else self._lazy.t, hip.weight_epoch(),
                tuple(p._version for p in self._tables()))

    def _bf16_arena_for_inference(self):
        a = self.__dict__.get("_arena_bf16")
        if a is None or torch.is_grad_enabled():
            return None
        stamp = self._bf16_stamp_now()
        if stamp != self._bf16_stamp or a.device != self._arena.device:
            raise RuntimeError("the bf16 snapshot of the embedding tables is stale (the model was trained or moved since "
                               "bf16_lookup()): call bf16_lookup() again, or bf16_lookup(False) to read the fp32 tables")
        return a

    def gather_linear(self, X, dense: List[torch.Tensor], linear: nn.Linear, out_link, pad_to: int = 64):
        """HIP path: (h1 [B, 64] = relu(linear(cat(emb, dense))), fm [B,1]) in one launch; x is written for the backward
        but never re-read in the forward."""
        self._ensure_packed()
        idx = self._idx_list(X)
        a16 = self._bf16_arena_for_inference()
        if a16 is not None:
            from ... import hip
            dense = [t.float().reshape(-1).contiguous() for t in dense]
            out = hip.embed_gather_linear_fwd_bf16(a16, self.row_base, self.row_count, idx, dense, Fh._rows16(linear.weight),
                                                   linear.bias, self.err_flag)
            if self.check_indices == "sync":
                self.raise_if_bad_index()
            return out
        F, D = len(idx), self.embedding_dim
        d = F * D + len(dense)
        ldx = (d + pad_to - 1) // pad_to * pad_to
        dense = [t if (t.dtype is torch.float32 and t.dim() == 1 and t.is_contiguous()) else t.float().reshape(-1).contiguous()
                 for t in dense]
        src = tuple(X[c] for c in self.emb_feature)
        self._presorted = None
        shadow = self._shadow_for_training()  # (before the replay below: a rebuild flushes)
        if self._lazy is not None and self._lazy.t > 0:
            keys, sk, sp = self._sorted_keys(idx, self.row_base, self.row_count, src)
            self._replay_rows(sk)
            self._presorted = (keys, sk, sp)
        elif torch.is_grad_enabled():
            self._presorted = self._sorted_keys(idx, self.row_base, self.row_count, src, lookup_only=True)
        out = Fh.embed_gather_linear(self, idx, dense, ldx, linear.weight, linear.bias, out_link, shadow=shadow)
        self._start_sort_ahead()
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return out

    def _replay_rows(self, sk) -> None:
        """the rows of this batch are brought up to date before they are read (LazyAdamRows.replay) — unless the step before
        this one already did that for exactly this key list (catch_up_ahead)"""
        done, self._ahead_done = self.__dict__.get("_ahead_done"), None
        if done is not None and done is sk:
            return
        self._lazy.replay(self, sk)

    def catch_up_ahead(self, sk) -> bool:
        """graph_step's "catch-up ahead": the optimizer catch-up of the NEXT batch's rows (`sk`: its sorted keys, the very
        tensor its forward will find), issued at the end of the step in progress — after the optimizer step, beside the dense
        Adam launch.  Same launch, same inputs, earlier: a second catch-up of rows that are already stamped leaves them alone
        (csrc/adam.hip "DEFERRED execution"), so a forward that does not find this promise just runs its own."""
        lz = self._lazy
        if lz is None or not lz.defer or lz.t <= 0:
            return False
        with torch.enable_grad():  # (stamps the rows for the step whose backward will write their gradients)
            lz.replay(self, sk)
        self._ahead_done = sk
        return True

    def _sorted_keys(self, idx, row_base, row_count, src, lookup_only: bool = False):
        """(keys, sorted keys, positions) of this batch's row requests.  Two layers with the same vocabularies fed the
        same batch (a model's D-wide tables and its LR_Layer's 1-wide ones) ask for the same arena rows: the second one
        reuses the first one's sort.  The cache entry keeps the id tensors alive, so `is` + `_version` identify them."""
        from ... import hip
        sig = (self._rows_sig(), str(self._arena.device))
        if src is not None:
            for c_src, c_sig, c_out in _SORT_PINNED:
                if c_sig == sig and len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src)):
                    return c_out
            for entry in _SORT_CACHE:
                c_src, c_ver, c_sig, c_out, c_event = entry
                if c_sig == sig and len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src)) \
                        and c_ver == tuple(t._version for t in src):
                    if c_event is not None:  # sorted ahead on the side stream: order this stream behind it
                        cur = torch.cuda.current_stream(self._arena.device)
                        cur.wait_event(c_event)
                        for t in c_out:
                            t.record_stream(cur)
                        entry[4] = None
                    return c_out
        if lookup_only:
            return None
        keys = hip.embed_keys(row_base, row_count, idx, self.err_flag)
        sk, sp = self._sort_pairs(keys) if row_base is self.row_base else hip.sort_pairs(keys, end_bit=self._meta()[3])
        if row_base is self.row_base and torch.is_grad_enabled():
            self._mark_sorted(sk, sp)
        out = (keys, sk, sp)
        if src is not None:
            self._cache_sort(src, sig, out, None)
        return out

    @staticmethod
    def _cache_sort(src, sig, out, event):
        _SORT_CACHE.append([src, tuple(t._version for t in src), sig, out, event])
        del _SORT_CACHE[:-4]

    def prefetch_sort(self, X) -> None:
        """Announce the batch of the NEXT step.  Its (arena row, position) sort is started on this layer's side stream
        right after this layer's next forward has been enqueued, so that it runs beside that step's backward and
        optimizer kernels: the radix sort is a chain of short latency-bound launches (0.17 ms at Criteo shape on a few %
        of the chip), the kernels beside it are HBM-bound.  `X` must hold the very tensors the later forward is given
        (identity is the cache key).  No effect on results: the same kernels on the same inputs, another stream."""
        if self._arena.is_cuda and all(X[c].device == self._arena.device for c in self.emb_feature):
            self._ahead = X

    def _start_sort_ahead(self) -> None:
        from ... import hip
        X, self._ahead = self.__dict__.get("_ahead"), None
        if X is None:
            return
        src = tuple(X[c] for c in self.emb_feature)
        sig = (self._rows_sig(), str(self._arena.device))
        pinned = next((c_out for c_src, c_sig, c_out in _SORT_PINNED
                       if c_sig == sig and len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src))), None)
        if pinned is not None:
            # static input buffers of a graphed step: re-sort into the persistent tensors, ON THIS STREAM — their readers
            # (a captured step's launches, eager forwards that match the pinned entry by identity) take no event from a
            # side stream (the side-stream form had no join back)
            self._sort_into(X, pinned, on_side_stream=False)
            return
        for c_src, c_ver, c_sig, _, _ in _SORT_CACHE:
            if c_sig == sig and len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src)) \
                    and c_ver == tuple(t._version for t in src):
                return
        dev = self._arena.device
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = hip.make_side_stream(dev)  # (a high-priority stream measured no better)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)  # the id tensors (and whatever produced them) are ordered on the caller's stream
        with torch.cuda.stream(side):
            idx = self._idx_list(X)
            # 'sync' mode raises at the step that finds a bad id: the keys of the NEXT batch must not set this step's flag
            # (its own gather checks the same ids again when its turn comes) — they get a scratch flag
            keys = hip.embed_keys(self.row_base, self.row_count, idx, self._ahead_flag())
            sk, sp = self._sort_pairs(keys)
            self._mark_sorted(sk, sp)
            event = torch.cuda.Event()
            event.record(side)
        for t in src:
            t.record_stream(side)
        self._cache_sort(src, sig, (keys, sk, sp), event)

    def _ahead_flag(self):
        if self.check_indices != "sync":
            return self.err_flag
        f = self.__dict__.get("_scratch_flag")
        if f is None or f.device != self._arena.device:
            f = self.__dict__["_scratch_flag"] = torch.zeros((1,), dtype=torch.int32, device=self._arena.device)
        return f

    def _sort_pairs(self, keys, out=None, workspace=None):
        """the (arena row, position) sort of a pair list that covers ALL fields of this layer in order, B pairs each: field
        segment by field segment (rp_sort_pairs_fields_i32: a table of r rows needs log2 r key bits, not the arena's — the same
        result in 49 instead of 78 field-passes at Criteo shape); RP_SORT_FIELDS=0 / RP_SORT=rocprim: the plain sort"""
        from ... import hip
        F = len(self.emb_feature)
        n = keys.numel()
        if n > 0 and n % F == 0 and os.environ.get("RP_SORT_FIELDS", "1") != "0" and os.environ.get("RP_SORT") != "rocprim":
            return hip.sort_pairs_fields(keys, n // F, self._rows_sig(), out=out, workspace=workspace)
        return hip.sort_pairs(keys, end_bit=self._meta()[3], out=out, workspace=workspace)

    def _sort_into(self, X, out, on_side_stream: bool) -> None:
        from ... import hip
        keys, sk, sp = out
        dev = self._arena.device
        if not on_side_stream:
            hip.embed_keys(self.row_base, self.row_count, self._idx_list(X), self.err_flag, out=keys)
            self._sort_pairs(keys, out=(sk, sp), workspace=_PINNED_WS.get(id(keys)))
            # (a pinned entry keeps the choice made when it was pinned: marks that first appear inside a recorded step's side
            #  section would be allocated by the capture behind the step's temporaries and run beside them)
            if _PINNED_MARKS.get(id(keys), True):
                self._mark_sorted(sk, sp, force=True)
            return
        side = _SIDE_STREAMS.get(dev)
        if side is None:
            side = _SIDE_STREAMS[dev] = hip.make_side_stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            hip.embed_keys(self.row_base, self.row_count, self._idx_list(X), self.err_flag, out=keys)
            self._sort_pairs(keys, out=(sk, sp))
            self._mark_sorted(sk, sp)

    def pin_sort(self, X) -> None:
        """graph_step: X holds STATIC id tensors (refilled in place from now on).  Sort their current content into
        persistent tensors on the current stream and pin the cache entry; later prefetch_sort(X) calls re-sort in place."""
        from ... import hip
        src = tuple(X[c] for c in self.emb_feature)
        sig = (self._rows_sig(), str(self._arena.device))
        for c_src, c_sig, c_out in _SORT_PINNED:
            if c_sig == sig and len(c_src) == len(src) and all(a is b for a, b in zip(c_src, src)):
                self._sort_into(X, c_out, on_side_stream=False)
                return
        n = sum(t.numel() for t in src)
        out = tuple(torch.empty((n,), dtype=torch.int32, device=self._arena.device) for _ in range(3))
        ws = hip.sort_workspace(n, self._arena.device)
        if n % len(src) == 0 and n > 0:  # (the field-segmented sort's histograms are per field: a few KB more)
            ws2 = hip.sort_fields_workspace(n // len(src), len(src), self._arena.device)
            ws = ws2 if ws2.numel() > ws.numel() else ws
        _PINNED_WS[id(out[0])] = ws
        _PINNED_MARKS[id(out[0])] = bool(getattr(self, "_marks_wanted", False))
        self._sort_into(X, out, on_side_stream=False)
        _SORT_PINNED.append((src, sig, out))

    @staticmethod
    def unpin_sorts(X=None) -> None:
        """drop the pinned entries of the static batch X (all of them when X is None)"""
        if X is None:
            del _SORT_PINNED[:]
            _PINNED_WS.clear()
            _PINNED_MARKS.clear()
        else:
            ids = {id(t) for t in X.values()}
            for e in _SORT_PINNED:
                if any(id(t) in ids for t in e[0]):
                    _PINNED_WS.pop(id(e[2][0]), None)
                    _PINNED_MARKS.pop(id(e[2][0]), None)
            _SORT_PINNED[:] = [e for e in _SORT_PINNED if not any(id(t) in ids for t in e[0])]

    def _rows_sig(self):
        if getattr(self, "_rows_sig_cache", None) is None:
            self._rows_sig_cache = tuple(int(t.shape[0]) for t in self._tables())
        return self._rows_sig_cache

    def _gather(self, idx, meta, dense, want_fm: bool, pad_to: int, src=None):
        """idx: one contiguous int64 [n] per addressed table; meta = (row_base, row_count) of those tables or None
        for all fields in order; src: the batch's id tensors (identity = sort cache key) or None."""
        F, D = len(idx), self.embedding_dim
        d = F * D + len(dense)
        ldx = (d + pad_to - 1) // pad_to * pad_to
        dense = [t.float().reshape(-1).contiguous() for t in dense]
        row_base, row_count = meta if meta is not None else (self.row_base, self.row_count)
        self._presorted = None
        if self._lazy is not None and self._lazy.t > 0:
            # exact lazy dense Adam: the rows this batch reads must first replay the zero-gradient steps they
            # skipped.  The (row, position) sort the backward needs anyway is done here and reused there.
            keys, sk, sp = self._sorted_keys(idx, row_base, row_count, src if meta is None else None)
            self._replay_rows(sk)
            self._presorted = (keys, sk, sp)
        elif src is not None and meta is None and torch.is_grad_enabled():
            # no replay to do (dense Adam / first step): the backward sorts — unless another layer already has
            self._presorted = self._sorted_keys(idx, row_base, row_count, src, lookup_only=True)
        out = Fh.embed_gather(self, idx, dense, ldx, want_fm, meta)
        self._start_sort_ahead()
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return out if want_fm else (out, None)

    def table_range(self, field: int):
        """(first arena row, rows) of table `field` as python ints (cached with the row signature)"""
        sig = self._rows_sig()
        return sum(sig[:field]), sig[field]

    def lookup_pooled(self, X: Dict[str, torch.Tensor], name: str, pooling: str = "sum",
                      offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The multi-id lookup of a `_seq` column fused with its pooling: what the reference writes as
        `MaskedSumPooling()(emb(X, name=name))` / `MaskedAveragePooling()(...)` (embedding.py:64-71 +
        layers/sequence.py:13-59), without the [B, L, D] intermediate -> [B, D].

        X[name]: [B, L] ids (dense bags, padding ids included, as the reference takes them) or, with `offsets`
        (int64 [B + 1]), a flat [nnz] id list in CSR form (ragged bags).  pooling: "sum" | "average" (the masked average
        counts non-zero ELEMENTS per (sample, column), like the reference)."""
        if pooling not in ("sum", "average"):
            raise ValueError("pooling must be 'sum' or 'average'")
        self._ensure_packed()
        base_name = name.replace("_seq", "") if "seq" in name else name
        ids = X[name].long()
        if not self._arena.is_cuda:  # BASELINE config 0 (CPU plumbing): the reference's own composition
            table = self.embedding_layer[base_name]
            if offsets is None:
                e = table(ids if ids.dim() == 2 else ids.view(-1, 1))
                s = e.sum(dim=1)
                return s if pooling == "sum" else s / (e.ne(0).sum(dim=1).to(e.dtype) + 1e-16)
            rows = table(ids.view(-1))  # ragged bags: segment sums over the flat id list
            lens = offsets[1:] - offsets[:-1]
            bag = torch.repeat_interleave(torch.arange(lens.numel()), lens)
            s = torch.zeros((lens.numel(), rows.shape[1]), dtype=rows.dtype).index_add(0, bag, rows)
            if pooling == "sum":
                return s
            cnt = torch.zeros_like(s).index_add(0, bag, (rows != 0).to(rows.dtype))
            return s / (cnt + 1e-16)
        f = self.emb_feature.index(base_name)
        if offsets is None:
            if ids.dim() != 2:
                ids = ids.view(-1, 1)
            B, L = ids.shape
        else:
            offsets = offsets.long().contiguous()
            B, L = offsets.numel() - 1, 0
        flat = ids.reshape(-1).contiguous()
        presorted = None
        if self._lazy is not None and self._lazy.t > 0:
            # lazy dense Adam: the rows this lookup reads first replay the zero-gradient steps they skipped
            from ... import hip
            keys = hip.embed_keys(self.row_base[f:f + 1], self.row_count[f:f + 1], [flat], self.err_flag)
            sk, sp = hip.sort_pairs(keys, end_bit=self._meta()[3])
            self._lazy.replay(self, sk)
            presorted = (sk, sp)
        out = Fh.embed_gather_pool(self, f, flat, offsets, L, B, pooling, presorted)
        if self.check_indices == "sync":
            self.raise_if_bad_index()
        return out

    def forward(self, X: Dict[str, torch.Tensor], name: Optional[str] = None) -> torch.Tensor:
        self._ensure_packed()
        if name is None:
            if self._arena.is_cuda:
                x, _ = self.gather_concat(X, [], want_fm=False, pad_to=1)
                return x.view(x.shape[0], len(self.emb_feature), self.embedding_dim)
            feature_emb_list = []
            for col in self.emb_feature:
                inp = X[col].long().view(-1, 1)
                feature_emb_list.append(self.embedding_layer[col](inp))
            return torch.stack(feature_emb_list, dim=1).squeeze(2)
        # single-field / sequence lookups (embedding.py:64-71): the same gather kernel over ONE table, every
        # (sample, position) of a `_seq` column being one row of the batch it sees
        base_name = name.replace("_seq", "") if "seq" in name else name
        if self._arena.is_cuda:
            i = self.emb_feature.index(base_name)
            ids = X[name].long()
            shape = tuple(ids.shape) if "seq" in name else (ids.numel(), 1)
            meta = (self.row_base[i:i + 1], self.row_count[i:i + 1])
            x, _ = self._gather([ids.reshape(-1).contiguous()], meta, [], False, 1)
            return x.view(*shape, self.embedding_dim)
        if "seq" in name:
            inp = X[name].long()
            fea = self.embedding_layer[name.replace("_seq", "")](inp)
        else:
            inp = X[name].long().view(-1, 1)
            fea = self.embedding_layer[name](inp)
        return fea
