"""rp_embed_grad_ss (round 6) alone at Criteo shape: the segment-sum launch and the matrix launch over the unique-row tiles,
each kernel's duration from rocprofv3 (run under `rocprofv3 --kernel-trace --stats`), with the marks made ahead.
    python profiles/microbench/probes/probe_grad_ss.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd import hip  # noqa: E402

dev = torch.device("cuda")
enc = bench.criteo_enc_dict(1)
B, D = int(os.environ.get("PROBE_B", "65536")), 64
fields = [k for k, v in enc.items() if "vocab_size" in v]
F = len(fields)
rows = [enc[c]["vocab_size"] + 1 for c in fields]
base = torch.tensor([sum(rows[:i]) for i in range(F)], dtype=torch.int64, device=dev)
cnt = torch.tensor(rows, dtype=torch.int64, device=dev)
R = sum(rows)
arena = torch.randn(R, D, device=dev)
G = torch.zeros(R, D, device=dev)
batch = bench.synth_batch(enc, B, 1, dev, os.environ.get("PROBE_DIST", "uniform"))
idx = [batch[c] for c in fields]
err = torch.zeros(1, dtype=torch.int32, device=dev)
keys = hip.embed_keys(base, cnt, idx, err)
sk, sp = hip.sort_pairs(keys, end_bit=int(R - 1).bit_length())
K = F * 64 + 13
dh = torch.randn(B, 64, device=dev) * (torch.rand(B, 64, device=dev) < 0.5)
W = torch.randn(64, K, device=dev) / K ** 0.5
gfm = torch.randn(B, device=dev)
ssum = torch.randn(B, D, device=dev)
tiny = [f for f in range(F) if rows[f] <= 254]
smp = [f for f in range(F) if rows[f] >= B]
skip = sum(1 << f for f in tiny + smp)
print("kept fields:", [(f, rows[f]) for f in range(F) if not (skip >> f) & 1])
dw = torch.zeros(64, K, device=dev)
marks = hip.embed_grad_ss_mark(sk, B, skip)


def timed(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:70s} {a.elapsed_time(b) / n:.4f} ms")


timed("embed_grad_ss_mark", lambda: hip.embed_grad_ss_mark(sk, B, skip, out=marks))
timed("embed_grad_ss (+ dw), marks made ahead", lambda: hip.embed_grad_ss(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip, field_rows=rows, dw=dw, marks=marks))
timed("embed_grad_ss without dw", lambda: hip.embed_grad_ss(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip, field_rows=rows, marks=marks))
timed("embed_grad_seg (+ dw) over the same fields", lambda: hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip, field_rows=rows, dw=dw))
