"""Checkpoints of row-sharded models in the REFERENCE layout, with optimizer state.

The reference saves `{'model': model.state_dict(), 'enc_dict': enc_dict}` to `<dir>/model.pth`
(rec_pangu/trainer.py:124-164) and its inference path does `model.load_state_dict(ckpt['model'])`
(examples/ranking/inference_example.py:29-37).  A model whose tables are row-sharded over G ranks
(rec_pangu_amd.sharded) has one `local_arena` per embedding layer instead of the reference's per-table
`<layer>.embedding_layer.<col>.weight` tensors, so its own state_dict() is not that layout.  This module writes
and reads the reference layout without ever materialising a full arena on a device:

  save_checkpoint   every rank writes `shard_<rank>_of_<world>.pth` (its local arenas, and — with an optimizer — its
                    Adam moments and step count; rank 0 adds the replicated dense parameters / buffers and their
                    moments); after a barrier rank 0 MERGES the shard files on the host, table by table, into
                      model.pth      {'model': <reference state_dict>, 'enc_dict': ...}          (what the reference writes)
                      optimizer.pth  {'step', 'param_groups', 'state': {<same keys>: {'exp_avg', 'exp_avg_sq'}}}
  load_checkpoint   any world size (re-shards): every rank memory-maps the merged files and copies the rows it owns
                    (global arena row r lives on rank r % G at local row r // G); single-process models load the very
                    same files — `model.load_state_dict(torch.load('model.pth')['model'])` is the reference's own path.

Resume is exact: weights, moments and the step count round-trip bit for bit (the lazily executed dense Adam is
flushed first, so every row is at the optimizer's last step), and a run that saves, reloads and continues equals
the uninterrupted run (tests/test_sharded_gloo.py, tests/test_hip_models.py).
"""
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.distributed as dist
from torch import nn


def _sharded_layers(model: nn.Module):
    from .sharded import ShardedEmbeddingLayer
    return OrderedDict((n, m) for n, m in model.named_modules() if isinstance(m, ShardedEmbeddingLayer))


def _dense_layers(model: nn.Module):
    from .models.layers.embedding import EmbeddingLayer
    return OrderedDict((n, m) for n, m in model.named_modules() if isinstance(m, EmbeddingLayer))


_SOLO = object()  # save_checkpoint(group=_SOLO): this process alone, whatever process groups exist


def _world(group):
    if group is _SOLO:
        return 1, 0
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


SQRT_KEY = "exp_avg_sq_sqrt"  # FusedAdam keeps sqrt(second moment) (rec_pangu_amd/optim.py); torch.optim.Adam the moment


def _table_state(optimizer, param):
    """{'exp_avg': ..., 'exp_avg_sq' | 'exp_avg_sq_sqrt': ...} of one parameter: the optimizer's NATIVE second-moment
    form is what gets saved, so that a resume with the same optimizer type is bit-exact; load converts when needed."""
    st = optimizer.state.get(param, None)
    if not st or "exp_avg" not in st:
        return None
    key = SQRT_KEY if SQRT_KEY in st else "exp_avg_sq"
    return {"exp_avg": st["exp_avg"].detach().cpu(), key: st[key].detach().cpu()}


def _step_of(optimizer) -> int:
    g = optimizer.param_groups[0]
    if "_rp_step" in g:
        return int(g["_rp_step"])
    for st in optimizer.state.values():
        if "step" in st:
            return int(st["step"])
    return 0


def shard_file(ckpt_dir: str, rank: int, world: int) -> str:
    return os.path.join(ckpt_dir, f"shard_{rank:03d}_of_{world:03d}.pth")


def _save_atomic(obj, path: str) -> None:
    """torch.save into a temporary file of the same directory, then os.replace: a reader (another rank that did not wait for
    the writer — save_checkpoint(collective=None) of a replicated model has no barrier) sees the old file or the
    complete new one, never a partial one"""
    tmp = f"{path}.tmp.{os.getpid()}"
    try:
        torch.save(obj, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def save_checkpoint(model: nn.Module, enc_dict: Optional[dict], ckpt_dir: str, optimizer=None, group=None,
                    merge: bool = True, filename: str = "model.pth", keep_shards: bool = False,
                    collective: Optional[bool] = None) -> None:
    """See the module docstring.

    * Model with row-SHARDED tables over more than one rank: always COLLECTIVE over `group` — EVERY rank must call it (two
      barriers inside); a caller that saves "on rank 0 only" would deadlock, and `collective=False` raises.
    * REPLICATED (unsharded) model in a multi-rank job: every rank holds the same weights and there is ONE writer, rank 0
      of `group`.  `collective=None` / `False` (the default): NO barrier — rank 0's call writes, any other rank's call only
      flushes its own optimizer state and returns, so both `if rank == 0: save_checkpoint(...)` and "every rank calls it"
      work and neither can hang.  `collective=True`: every rank calls it and all of them return only after the
      file is complete (one barrier).
    keep_shards=False: after a successful merge every rank removes its `shard_*` file (each holds a full shard of the
    weights and moments; `model.pth` / `optimizer.pth` are what `load_checkpoint` reads)."""
    world, rank = _world(group)
    layers = _sharded_layers(model)
    if layers and world > 1 and collective is False:
        raise ValueError("save_checkpoint: a model with row-sharded tables is saved collectively (every rank holds a shard)")
    if not layers:
        job_world, job_rank = world, rank
        world, rank = 1, 0
        if job_world > 1:
            try:
                if job_rank == 0:
                    save_checkpoint(model, enc_dict, ckpt_dir, optimizer, _SOLO, merge, filename, keep_shards)
                elif optimizer is not None and hasattr(optimizer, "flush"):
                    optimizer.flush()  # (every rank's state moves the same way as the writer's)
            finally:
                if collective:
                    dist.barrier(group=group)
            return
    os.makedirs(ckpt_dir, exist_ok=True, mode=0o777)
    if optimizer is not None and hasattr(optimizer, "flush"):
        optimizer.flush()  # lazy dense Adam: every row to the optimizer's last step
    sd = model.state_dict()  # (state_dict hooks flush too)
    arena_keys = {f"{n}.local_arena" if n else "local_arena" for n in layers}
    payload = {"rank": rank, "world": world, "layers": {}, "dense": None, "optimizer": None}
    for name, lay in layers.items():
        payload["layers"][name] = {"local_arena": lay.local_arena.detach().cpu(), "rows": list(lay._rows),
                                   "emb_feature": list(lay.emb_feature), "embedding_dim": lay.embedding_dim}
    if rank == 0:
        payload["dense"] = OrderedDict((k, v.detach().cpu()) for k, v in sd.items() if k not in arena_keys)
        payload["key_order"] = list(sd.keys())
    if optimizer is not None:
        named = dict(model.named_parameters())
        opt = {"step": _step_of(optimizer), "tables": {}, "dense": {},
               "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]}
        for name, lay in layers.items():
            ts = _table_state(optimizer, lay.local_arena)
            if ts is not None:
                opt["tables"][name] = ts
        if rank == 0:
            for k, p in named.items():
                if k in arena_keys:
                    continue
                ts = _table_state(optimizer, p)
                if ts is not None:
                    opt["dense"][k] = ts
        payload["optimizer"] = opt
    _save_atomic(payload, shard_file(ckpt_dir, rank, world))
    if world > 1:
        dist.barrier(group=group)
    if merge and rank == 0:
        merge_shards(ckpt_dir, world, enc_dict, filename)
    if world > 1:
        dist.barrier(group=group)
    if merge and not keep_shards:
        try:
            os.remove(shard_file(ckpt_dir, rank, world))
        except OSError:
            pass


def _merge_table(parts, f: int, rows, base, D: int, world: int, dtype):
    """Full [rows[f], D] tensor of table f from the ranks' local arenas (`parts[g]`: rank g's arena)."""
    out = torch.empty((rows[f], D), dtype=dtype)
    for g in range(world):
        first = (g - base[f]) % world
        dst = out[first::world]
        lrow = (base[f] + first) // world
        dst.copy_(parts[g][lrow:lrow + dst.shape[0]])
    return out


def merge_shards(ckpt_dir: str, world: int, enc_dict: Optional[dict], filename: str = "model.pth") -> None:
    """Host-side merge of `shard_*_of_<world>.pth` into model.pth (+ optimizer.pth) in the reference layout."""
    shards = [torch.load(shard_file(ckpt_dir, g, world), map_location="cpu", weights_only=False, mmap=True)
              for g in range(world)]
    head = shards[0]
    model_sd: Dict[str, torch.Tensor] = OrderedDict()
    opt_state: Dict[str, dict] = OrderedDict()
    has_opt = head["optimizer"] is not None
    for key in head["key_order"]:
        lname = key[:-len(".local_arena")] if key.endswith(".local_arena") else (
            "" if key == "local_arena" else None)
        if lname is None or lname not in head["layers"]:
            model_sd[key] = head["dense"][key]
            if has_opt and key in head["optimizer"]["dense"]:
                opt_state[key] = head["optimizer"]["dense"][key]
            continue
        info = head["layers"][lname]
        rows, D = info["rows"], info["embedding_dim"]
        base = [0]
        for r in rows[:-1]:
            base.append(base[-1] + r)
        parts = [s["layers"][lname]["local_arena"] for s in shards]
        mparts = vparts = vkey = None
        if has_opt and lname in head["optimizer"]["tables"]:
            vkey = SQRT_KEY if SQRT_KEY in head["optimizer"]["tables"][lname] else "exp_avg_sq"
            mparts = [s["optimizer"]["tables"][lname]["exp_avg"] for s in shards]
            vparts = [s["optimizer"]["tables"][lname][vkey] for s in shards]
        for f, col in enumerate(info["emb_feature"]):
            k = f"{lname}.embedding_layer.{col}.weight" if lname else f"embedding_layer.{col}.weight"
            model_sd[k] = _merge_table(parts, f, rows, base, D, world, parts[0].dtype)
            if mparts is not None:
                opt_state[k] = {"exp_avg": _merge_table(mparts, f, rows, base, D, world, mparts[0].dtype),
                                vkey: _merge_table(vparts, f, rows, base, D, world, vparts[0].dtype)}
    ckpt = {"model": model_sd}
    if enc_dict is not None:
        ckpt["enc_dict"] = enc_dict
    _save_atomic(ckpt, os.path.join(ckpt_dir, filename))
    if has_opt:
        _save_atomic({"step": head["optimizer"]["step"], "param_groups": head["optimizer"]["param_groups"],
                    "state": opt_state}, os.path.join(ckpt_dir, "optimizer.pth"))


def _second_moment(saved: dict, want_sqrt: bool):
    """the saved second moment in the form the loading optimizer keeps (exact when the forms agree)"""
    if want_sqrt:
        return saved[SQRT_KEY] if SQRT_KEY in saved else saved["exp_avg_sq"].sqrt()
    return saved["exp_avg_sq"] if "exp_avg_sq" in saved else saved[SQRT_KEY] * saved[SQRT_KEY]


def _install_moments(optimizer, param, exp_avg, second, step: int):
    from .optim import FusedAdam
    st = optimizer.state[param]
    st["exp_avg"] = exp_avg.to(device=param.device, dtype=param.dtype).clone()
    if isinstance(optimizer, FusedAdam):
        st.pop("exp_avg_sq", None)
        st[SQRT_KEY] = second.to(device=param.device, dtype=param.dtype).clone()
    else:  # torch.optim.Adam keeps the moment itself and a per-parameter step tensor
        st["exp_avg_sq"] = second.to(device=param.device, dtype=param.dtype).clone()
        st["step"] = torch.tensor(float(step))


def load_checkpoint(model: nn.Module, ckpt_dir: str, optimizer=None, group=None, filename: str = "model.pth") -> dict:
    """Load model.pth (+ optimizer.pth when `optimizer` is given) written by save_checkpoint — or by the reference's
    own save_all — into `model`, re-sharding the tables for the current world size.  Returns the checkpoint's
    non-tensor entries ({'enc_dict': ...} when present)."""
    world, rank = _world(group)
    layers = _sharded_layers(model)
    ckpt = torch.load(os.path.join(ckpt_dir, filename), map_location="cpu", weights_only=False, mmap=True)
    full = ckpt["model"]
    opt_ck = None
    if optimizer is not None:
        opt_ck = torch.load(os.path.join(ckpt_dir, "optimizer.pth"), map_location="cpu", weights_only=False, mmap=True)
        if hasattr(optimizer, "flush"):
            optimizer.flush()
    if not layers:
        model.load_state_dict(full)
    else:
        arena_keys = {f"{n}.local_arena" if n else "local_arena" for n in layers}
        dense = OrderedDict((k, v) for k, v in full.items()
                            if not any(k.startswith((n + "." if n else "") + "embedding_layer.") for n in layers))
        missing, unexpected = model.load_state_dict(dense, strict=False)
        bad = [k for k in missing if k not in arena_keys]
        if bad or unexpected:
            raise RuntimeError(f"load_checkpoint: missing {bad}, unexpected {list(unexpected)}")
        with torch.no_grad():
            for name, lay in layers.items():
                lay.flush_lazy()
                for f, col in enumerate(lay.emb_feature):
                    src = full[f"{name}.embedding_layer.{col}.weight" if name else f"embedding_layer.{col}.weight"]
                    if tuple(src.shape) != (lay._rows[f], lay.embedding_dim):
                        raise RuntimeError(f"load_checkpoint: table {name}/{col} is {tuple(src.shape)}, "
                                           f"the model expects {(lay._rows[f], lay.embedding_dim)}")
                    first, lrow = lay._table_slice(f)
                    mine = src[first::lay.world]
                    lay.local_arena.data[lrow:lrow + mine.shape[0]].copy_(mine)
    if optimizer is not None:
        step = int(opt_ck["step"])
        for g, saved in zip(optimizer.param_groups, opt_ck["param_groups"]):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
            g["_rp_step"] = step
        if hasattr(optimizer, "_plans"):
            # FusedAdam: arena-backed moments are rebuilt from optimizer.state at the next step (_adopt_loaded_state)
            for store in list(getattr(optimizer, "_stores", {}).values()):
                store._lazy = None
            optimizer._arena_state.clear()
            optimizer._plans.clear()
        named = dict(model.named_parameters())
        want_sqrt = hasattr(optimizer, "SQRT_KEY")
        for k, p in named.items():
            lname = k[:-len(".local_arena")] if k.endswith(".local_arena") else None
            if lname is not None and lname in layers:
                lay = layers[lname]
                m = torch.zeros_like(lay.local_arena.data)
                v = torch.zeros_like(lay.local_arena.data)
                any_state = False
                for f, col in enumerate(lay.emb_feature):
                    st = opt_ck["state"].get(f"{lname}.embedding_layer.{col}.weight" if lname else
                                             f"embedding_layer.{col}.weight")
                    if st is None:
                        continue
                    any_state = True
                    first, lrow = lay._table_slice(f)
                    sm, sv = st["exp_avg"][first::lay.world], _second_moment(st, want_sqrt)[first::lay.world]
                    m[lrow:lrow + sm.shape[0]].copy_(sm)
                    v[lrow:lrow + sv.shape[0]].copy_(sv)
                if any_state:
                    _install_moments(optimizer, p, m, v, step)
                    if getattr(lay, "_lazy", None) is not None:
                        lay._lazy = None
            elif k in opt_ck["state"]:
                st = opt_ck["state"][k]
                _install_moments(optimizer, p, st["exp_avg"], _second_moment(st, want_sqrt), step)
    return {k: v for k, v in ckpt.items() if k != "model"}
