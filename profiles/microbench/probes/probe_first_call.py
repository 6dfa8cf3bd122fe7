"""The first replay behind a synchronisation: one hipLaunchKernel of it blocks 0.6-1.5 ms (bench.py host_stall).  Which one, and
does the kind of wait in front of it matter?  (profiles/r06_window_ramp.txt)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from rec_pangu_amd import hip
from rec_pangu_amd.graph_step import GraphedTrainStep
from rec_pangu_amd.optim import make_adam
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
B = 65536
g = GraphedTrainStep(model, opt)
gen = lambda i: bench.synth_batch(enc, B, 100 + i, dev)
nb = gen(0)
for i in range(300):
    cur, nb = nb, gen(i + 1)
    g(cur, nb)
torch.cuda.synchronize()
bs = [gen(5000 + i) for i in range(64)]
ctr = [0]
def run(n):
    for _ in range(n):
        i = ctr[0]; ctr[0] += 1
        g(bs[i % 64], bs[(i + 1) % 64])
tiny = torch.zeros(4, dtype=torch.int64, device=dev)
def wait(kind):
    if kind == "device_sync":
        torch.cuda.synchronize()
    elif kind == "stream_sync":
        torch.cuda.current_stream().synchronize()
    elif kind == "event_poll":
        ev = torch.cuda.Event(); ev.record()
        while not ev.query():
            pass
    elif kind == "sync_sleep2ms":
        torch.cuda.synchronize(); time.sleep(0.002)
    elif kind == "sync_then_tiny":
        torch.cuda.synchronize(); hip.counter_add(tiny, 1)
    elif kind == "sync_then_3tiny":
        torch.cuda.synchronize()
        for _ in range(3):
            hip.counter_add(tiny, 1)
KINDS = sys.argv[1:] or ["device_sync", "stream_sync", "event_poll", "sync_sleep2ms", "sync_then_tiny", "sync_then_3tiny", "device_sync"]
for kind in KINDS:
    res = []
    for rep in range(8):
        if kind.startswith("fresh"):
            # as bench.py does it: 256 replays on batches of their own, which are then dropped right in front of the wait
            n = int("".join(ch for ch in kind if ch.isdigit()) or 256)
            pw = [gen(900000 + rep * 1000 + i) for i in range(n)]
            for i in range(n - 1):
                g(pw[i], pw[i + 1])
            g(pw[n - 1], bs[ctr[0] % 64])
            if "keep" not in kind:
                del pw
            if "then40" in kind:
                run(40)
            torch.cuda.synchronize()
        else:
            run(40)
            wait(kind)
        for pl in g.plans:
            pl.slowest_call(reset=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        run(1)
        h = (time.perf_counter() - t0) * 1e3
        e1.record()
        sc = max((pl.slowest_call(reset=True) for pl in g.plans), key=lambda t_: t_[2])
        run(3)
        torch.cuda.synchronize()
        res.append((round(h, 3), sc[0], sc[1], round(sc[2], 3), round(e0.elapsed_time(e1), 3)))
    print(kind, res, flush=True)
