"""Eager step vs replays of the captured step (launch plan / hipGraph) at the headline shape: wall time per step with the
device kept busy, nothing else.  Under `rocprofv3 --kernel-trace` the CSV of the run shows where the device time goes
(profiles/trace_gaps.py).
    python profiles/microbench/probes/probe_plan.py eager|plan|hipgraph [steps] [batch]"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd.graph_step import GraphedTrainStep  # noqa: E402
from rec_pangu_amd.optim import make_adam  # noqa: E402

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
dev = torch.device("cuda")
enc = bench.criteo_enc_dict(1)
torch.manual_seed(0)
with torch.device(dev):
    model = bench.build_model("deepfm", enc)
for m in model.modules():
    if hasattr(m, "check_indices"):
        m.check_indices = "deferred"
model.train()
opt = make_adam(model, 1e-3)
batches = [bench.synth_batch(enc, B, 100 + i, dev) for i in range(32)]
g = None if mode == "eager" else GraphedTrainStep(model, opt, backend=mode)


def step(i):
    cur, nxt = batches[i % 32], batches[(i + 1) % 32]
    if g is not None:
        g(cur, nxt)
    else:
        model.prefetch(nxt)
        out = model(cur)
        out["loss"].backward()
        opt.step()
        model.zero_grad()


for i in range(40):
    step(i)
torch.cuda.synchronize()
done, parts = 40, []
for seg in ([steps] if steps <= 200 else [60, 140, 300, 600, steps - 1100] if steps > 1100 else [steps]):
    if seg <= 0:
        continue
    t0 = time.perf_counter()
    for i in range(done, done + seg):
        step(i)
    torch.cuda.synchronize()
    parts.append(f"steps {done}-{done + seg}: {(time.perf_counter() - t0) / seg * 1e3:.4f}")
    done += seg
print(f"{mode} at B={B}, ms/step: " + "; ".join(parts) + ("" if g is None else f" (backend used: {g.backend_used}, {g.why_not_plan})"))
