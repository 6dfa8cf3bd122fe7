"""FusedAdam — the optimiser RankTrainer.fit builds (reference: rec_pangu/trainer.py:75,
torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)), as fused HIP launches.

Semantics are the reference's: DENSE Adam over every parameter, embedding tables included — rows
nobody looked up still decay their moments and move.  Arena-backed embedding tables
(rec_pangu_amd.models.layers.EmbeddingLayer) are updated as ONE flat tensor per layer
(p/g/m/v arenas), all other parameters in one multi-tensor launch.  `fuse_zero_grad=True` clears
the gradients inside the same pass (the model.zero_grad() that follows optimizer.step() in the
reference loop, model_pipeline.py:57-58) so the 8.6 GB gradient arena is not streamed twice.
CPU parameters are not handled here: build torch.optim.Adam for a CPU model (make_adam does).
"""
from typing import Dict, List

import torch

from . import hip


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, fuse_zero_grad=False):
        if weight_decay != 0:
            raise ValueError("FusedAdam mirrors the reference's optimiser: weight_decay must be 0")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0))
        self.fuse_zero_grad = fuse_zero_grad
        self._arena_state: Dict[int, dict] = {}

    @staticmethod
    def _store_of(p):
        ref = getattr(p, "_rp_store", None)
        return None if ref is None else ref()

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
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam only updates HIP-device parameters (use make_adam for CPU models)")
                store = self._store_of(p)
                if store is not None:
                    sid = id(store)
                    if sid not in arena_ok:  # one check per layer, not per table
                        arena_ok[sid] = store.grads_are_arena()
                    if arena_ok[sid]:
                        stores[sid] = store
                        continue
                st = self.state[p]
                if not st:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                ps.append(p.data)
                gs.append(p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            for sid, store in stores.items():
                st = self._arena_state.get(sid)
                if st is None or st["m"].shape != store.arena.shape or st["m"].device != store.arena.device:
                    st = {"m": torch.zeros_like(store.arena), "v": torch.zeros_like(store.arena)}
                    self._arena_state[sid] = st
                    off = 0
                    for p in store.table_parameters():  # per-table views, for state_dict()/inspection
                        r = p.shape[0]
                        self.state[p]["exp_avg"] = st["m"][off:off + r]
                        self.state[p]["exp_avg_sq"] = st["v"][off:off + r]
                        off += r
                ps.append(store.arena.view(-1))
                gs.append(store.grad_arena.view(-1))
                ms.append(st["m"].view(-1))
                vs.append(st["v"].view(-1))
            if ps:
                hip.adam_step(ps, gs, ms, vs, lr, b1, b2, eps, step, self.fuse_zero_grad)
            if self.fuse_zero_grad:
                for store in stores.values():
                    store.grads_were_zeroed()
        return loss


def make_adam(model, lr):
    """What RankTrainer.fit uses: fused HIP Adam for a HIP-resident model, torch.optim.Adam on CPU
    (BASELINE config 0).  Hyper-parameters are the reference's (trainer.py:75)."""
    params = list(model.parameters())
    if params and params[0].is_cuda:
        return FusedAdam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0, fuse_zero_grad=True)
    return torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999), eps=1e-08, weight_decay=0)
