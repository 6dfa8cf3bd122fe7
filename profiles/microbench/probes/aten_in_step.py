"""Which launches of a training step are NOT the library's (ATen kernels, memsets, copies)?  They are what keeps a model's
captured step from replaying as a launch plan (graph_step.GraphedTrainStep.why_not_plan).  One eager step per model under
torch.profiler; prints every device activity whose name is not one of librecpangu_hip.so's kernels."""
import os
import sys
import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd.optim import make_adam  # noqa: E402

dev = torch.device("cuda")
B = 8192
for name in sys.argv[1:] or ["dcn", "mmoe", "autoint", "xdeepfm", "deepfm"]:
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
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(3)
        torch.cuda.synchronize()
    foreign = {}
    ours = 0
    for ev in prof.events():
        if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
            nm = ev.name
            if ("at::" in nm or "elementwise" in nm or "Memcpy" in nm or "Memset" in nm or "copyBuffer" in nm or "fillBuffer" in nm
                    or "rocprim" in nm or "cub::" in nm or "reduce_kernel" in nm and "at" in nm):
                foreign[nm[:110]] = foreign.get(nm[:110], 0) + 1
            else:
                ours += 1
    print(f"== {name}: {ours} library launches, {sum(foreign.values())} foreign")
    for k, v in sorted(foreign.items(), key=lambda kv: -kv[1]):
        print(f"   {v:3d} x {k}")
