"""The ATen launches of a ROW-SHARDED DeepFM training step (1-rank RCCL group, fixed-capacity exchange, deferred checks):
what keeps the recorded step from replaying as a launch plan in segments.  As aten_sources.py."""
import collections
import os
import sys
import traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd.optim import make_adam  # noqa: E402
from rec_pangu_amd.sharded import build_sharded_model, allreduce_dense_grads  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
VIEWS = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice",
         "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.unflatten",
         "aten.flatten", "aten.unbind", "aten.split", "aten.narrow", "aten.empty", "aten.empty_like", "aten.empty_strided",
         "aten.new_empty", "aten._local_scalar_dense", "aten.is_", "aten.size", "aten.stride", "aten.sym_", "aten.lift_fresh",
         "aten.result_type", "aten.can_cast", "aten.chunk", "aten.view_as", "aten._reshape_alias", "aten.resize_",
         "aten.set_", "aten.item", "aten.record_stream", "aten.is_pinned", "prim.")
B = 4096
enc = bench.criteo_enc_dict(64)
model = build_sharded_model(lambda: bench.build_model("deepfm", enc), 1, 0, dev, seed=0)
model.train()
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
opt = make_adam(model, 1e-3)
batches = [bench.synth_batch(enc, B, 7 + i, dev) for i in range(4)]


def step(i):
    out = model(batches[i % 4])
    out["loss"].backward()
    allreduce_dense_grads(model)
    opt.step()
    model.zero_grad()


for i in range(3):
    step(i)
seen = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        s = str(func)
        if not any(s.startswith(v) for v in VIEWS):
            dev_t = any(torch.is_tensor(a) and a.is_cuda for a in args) or \
                any(isinstance(a, (list, tuple)) and any(torch.is_tensor(x) and x.is_cuda for x in a) for a in args) or \
                ("device" in (kwargs or {}) and "cuda" in str((kwargs or {})["device"]))
            if dev_t:
                where = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if ("rec_pangu_amd" in fr.filename or "bench.py" in fr.filename) and "aten_sources" not in fr.filename:
                        where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70] if fr.line else ''}"
                        break
                seen[(s, where)] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    step(3)
torch.cuda.synchronize()
for (op, where), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {op:40s} {where}")
dist.destroy_process_group()
