"""Where do the ATen launches of a training step come from?  One eager step per model under a TorchDispatchMode (forward and
autograd thread): every dispatched op that is not a pure view is listed with the innermost rec_pangu_amd / optimizer source
line that issued it.  What is listed here is what keeps a captured step from replaying as a launch plan."""
import collections
import os
import sys
import traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd.optim import make_adam  # noqa: E402

VIEWS = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice",
         "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.unflatten",
         "aten.flatten", "aten.unbind", "aten.split", "aten.narrow", "aten.empty", "aten.empty_like", "aten.empty_strided",
         "aten.new_empty", "aten._local_scalar_dense", "aten.is_", "aten.size", "aten.stride", "aten.sym_", "aten.lift_fresh",
         "aten.result_type", "aten.can_cast", "aten.chunk", "aten.view_as", "aten._reshape_alias", "aten.resize_",
         "aten.set_", "aten.item", "aten.record_stream", "aten.is_pinned", "prim.", "aten.zeros.default" if False else "~")
dev = torch.device("cuda")
B = 4096
for name in sys.argv[1:] or ["mmoe", "autoint", "xdeepfm", "dcn", "deepfm"]:
    enc = bench.mmoe_enc_dict(64) if name == "mmoe" else bench.criteo_enc_dict(64)
    torch.manual_seed(0)
    with torch.device(dev):
        model = bench.build_model(name, enc)
    model.train()
    for m in model.modules():
        if hasattr(m, "check_indices"):
            m.check_indices = "deferred"
    opt = make_adam(model, 1e-3)
    batches = [bench.synth_batch(enc, B, 7 + i, dev) for i in range(4)]

    def step(i):
        model.prefetch(batches[(i + 1) % 4])
        out = model(batches[i % 4])
        out["loss"].backward()
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
    print(f"== {name}: {sum(seen.values())} non-view ATen dispatches on device tensors in one step")
    for (op, where), n in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"   {n:3d} x {op:38s} {where}")
