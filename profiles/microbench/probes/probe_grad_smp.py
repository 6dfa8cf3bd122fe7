"""rp_embed_grad_smp (round 6: the big tables' share of the first layer's backward, sample-major) + rp_embed_grad_seg over the
remaining fields against round 5's rp_embed_grad_seg over every non-tiny field, launches alone, back to back, at Criteo
shape.  PROBE_SMP_MIN = smallest table (rows) that goes sample-major (default: B).  Run on the GPU box:
    python profiles/microbench/probes/probe_grad_smp.py"""
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
batch = bench.synth_batch(enc, B, 1, dev)
idx = [batch[c] for c in fields]
err = torch.zeros(1, dtype=torch.int32, device=dev)
keys = hip.embed_keys(base, cnt, idx, err)
sk, sp = hip.sort_pairs(keys, end_bit=int(R - 1).bit_length())
ND = 13
K = F * 64 + ND
ldx = (K + 63) // 64 * 64
dh = torch.randn(B, 64, device=dev) * (torch.rand(B, 64, device=dev) < 0.5)
W = torch.randn(64, K, device=dev) / K ** 0.5
wt = hip.transpose(W, rows_out=ldx)
gfm = torch.randn(B, device=dev)
ssum = torch.randn(B, D, device=dev)
tiny = [(f, int(base[f]), rows[f]) for f in range(F) if rows[f] <= 254]
tot, pick = 0, []
for t in sorted(tiny, key=lambda t: t[2]):
    if tot + t[2] <= 224 and len(pick) < 16:
        pick.append(t)
        tot += t[2]
tiny = sorted(pick)
skip_tiny = sum(1 << t[0] for t in tiny)
smp_min = int(os.environ.get("PROBE_SMP_MIN", str(B)))
smp = [f for f in range(F) if rows[f] >= smp_min and not (skip_tiny >> f) & 1][:16]
skip_smp = sum(1 << f for f in smp)
smp_t = [(f, int(base[f]), rows[f]) for f in smp]
print(f"B={B} F={F} tiny={[t[0] for t in tiny]} smp={smp} rows={[rows[f] for f in smp]}")


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
    ms = a.elapsed_time(b) / n
    print(f"{name:78s} {ms:.4f} ms")
    return ms


dw = torch.zeros(64, K, device=dev)
marks = hip.embed_grad_smp_mark(sk, sp, B, smp)
ndup = int((marks[1] >= 0).sum())
print(f"duplicate pairs in the sample-major fields: {ndup} of {len(smp) * B}")
timed("r5: embed_grad_seg over every non-tiny field (+ dw)", lambda: hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip_tiny, field_rows=rows, dw=dw))
timed("embed_grad_smp_mark", lambda: hip.embed_grad_smp_mark(sk, sp, B, smp, out=marks))
timed("embed_grad_smp (+ dw partial sum + duplicate reduce)", lambda: hip.embed_grad_smp(keys, marks, B, F, smp_t, dh, W, gfm, ssum, arena, G, False, dw=dw))
timed("embed_grad_smp without dw", lambda: hip.embed_grad_smp(keys, marks, B, F, smp_t, dh, W, gfm, ssum, arena, G, False))
timed("embed_grad_seg over the remaining fields (+ dw)", lambda: hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip_tiny | skip_smp, field_rows=rows, dw=dw))
timed("embed_grad_ss over the remaining fields (+ dw)", lambda: hip.embed_grad_ss(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip_tiny | skip_smp, field_rows=rows, dw=dw))
timed("embed_grad_ss over the remaining AND the tiny fields (+ dw)", lambda: hip.embed_grad_ss(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip_smp, field_rows=rows, dw=dw))
timed("embed_grad_tiny (+ dw)", lambda: hip.embed_grad_tiny(keys, B, tiny, dh, wt, gfm, ssum, arena, G, False, dw=dw))


def both():
    hip.embed_grad_smp(keys, marks, B, F, smp_t, dh, W, gfm, ssum, arena, G, False, dw=dw)
    hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip_tiny | skip_smp, field_rows=rows, dw=dw)


timed("smp + seg back to back", both)
# correctness at full size: against r5's seg over every non-tiny field
Ga = torch.zeros(R, D, device=dev)
Gb = torch.zeros(R, D, device=dev)
dwa, dwb = torch.zeros(64, K, device=dev), torch.zeros(64, K, device=dev)
hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, Ga, False, skip_fields=skip_tiny, field_rows=rows, dw=dwa)
hip.embed_grad_smp(keys, marks, B, F, smp_t, dh, W, gfm, ssum, arena, Gb, False, dw=dwb)
hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, Gb, False, skip_fields=skip_tiny | skip_smp, field_rows=rows, dw=dwb)
torch.cuda.synchronize()
print("max |G_smp+seg - G_seg| =", float((Ga - Gb).abs().max()), " scale", float(Ga.abs().max()))
print("max |dw - dw| =", float((dwa - dwb)[:, :F * 64].abs().max()), " scale", float(dwa[:, :F * 64].abs().max()))

Gc = torch.zeros(R, D, device=dev)
dwc = torch.zeros(64, K, device=dev)
hip.embed_grad_smp(keys, marks, B, F, smp_t, dh, W, gfm, ssum, arena, Gc, False, dw=dwc)
hip.embed_grad_ss(sk, sp, B, D, dh, W, gfm, ssum, arena, Gc, False, skip_fields=skip_tiny | skip_smp, field_rows=rows, dw=dwc)
torch.cuda.synchronize()
print("max |G_smp+ss - G_seg| =", float((Ga - Gc).abs().max()))
print("max |dw_ss - dw| =", float((dwa - dwc)[:, :F * 64].abs().max()))
