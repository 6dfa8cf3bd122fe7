"""rp_embed_grad_seg (round 5: segment sums first, the first layer's weight gradient folded in, no stored activation)
against the round-4 form (rp_embed_grad_gemm + rp_embed_grad_tiny + rp_linear_wgrad over the stored x) at Criteo shape,
launches alone, back to back.  RP_SEG_ROWS / RP_SEG_TILES select the tile size / tiles per chunk (read once per process:
run the script once per setting).  Run on the GPU box:
    python profiles/microbench/probes/probe_grad_seg.py"""
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
x = torch.randn(B, ldx, device=dev)
xd = torch.zeros(B, 64, device=dev)
xd[:, :ND] = x[:, F * 64:K]
tiny = [(f, int(base[f]), rows[f]) for f in range(F) if rows[f] <= 254]
tot = 0
pick = []
for t in sorted(tiny, key=lambda t: t[2]):
    if tot + t[2] <= 224 and len(pick) < 16:
        pick.append(t)
        tot += t[2]
tiny = sorted(pick)
skip = sum(1 << t[0] for t in tiny)
uniq = int(torch.unique_consecutive(sk).numel())
print(f"B={B} F={F} pairs={F * B} unique rows={uniq} tiny tables={len(tiny)} ({tot} rows) "
      f"RP_SEG_ROWS={os.environ.get('RP_SEG_ROWS', '128')} RP_SEG_TILES={os.environ.get('RP_SEG_TILES', 'auto')}")


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


dw_new = torch.zeros(64, K, device=dev)


def old_gemm():
    hip.embed_grad_gemm(sk, sp, B, D, dh, wt, None, gfm, ssum, arena, G, False, skip_fields=skip)


def old_tiny():
    hip.embed_grad_tiny(keys, B, tiny, dh, wt, gfm, ssum, arena, G, False)


def old_wgrad():
    return hip.linear_wgrad(dh, x, K)


def new_seg():
    hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip, field_rows=rows, dw=dw_new)


def new_seg_nodw():
    hip.embed_grad_seg(sk, sp, B, D, dh, W, gfm, ssum, arena, G, False, skip_fields=skip, field_rows=rows)


def new_tiny():
    hip.embed_grad_tiny(keys, B, tiny, dh, wt, gfm, ssum, arena, G, False, dw=dw_new)


def new_dense():
    hip.linear_wgrad(dh, xd, ND, dw=dw_new[:, F * 64:])


a = timed("round 4: embed_grad_gemm (18 fields)", old_gemm)
b = timed("round 4: embed_grad_tiny (8 tables)", old_tiny)
c = timed("round 4: linear_wgrad over the stored x [B, 1728]", old_wgrad)
print(f"{'round 4 back to back':78s} {a + b + c:.4f} ms")
d = timed("round 5: embed_grad_seg (18 fields, table rows + dW columns)", new_seg)
timed("round 5: embed_grad_seg without the weight gradient", new_seg_nodw)
e = timed("round 5: embed_grad_tiny + its dW columns", new_tiny)
f_ = timed("round 5: linear_wgrad over xd (13 dense columns + bias)", new_dense)
print(f"{'round 5 back to back':78s} {d + e + f_:.4f} ms")

# agreement of the two forms on this data
G.zero_()
old_gemm(); old_tiny()
G_old = G.clone()
dw_old, _ = old_wgrad()
G.zero_()
new_seg(); new_tiny(); new_dense()
torch.cuda.synchronize()
gs = float(G_old.abs().max())
print(f"max |G_new - G_old| / max|G| = {float((G - G_old).abs().max()) / gs:.3e}")
xg = torch.cat([arena[keys[f * B:(f + 1) * B].long()] for f in range(F)], dim=1)
dw_ref = (dh.double().T @ torch.cat([xg, xd[:, :ND]], dim=1).double())
ws = float(dw_ref.abs().max())
print(f"max |dW_new - fp64| / max|dW| = {float((dw_new.double() - dw_ref).abs().max()) / ws:.3e}   "
      f"(stored-x GEMM on random x: not comparable; dense columns: {float((dw_new[:, F*64:].double() - dw_ref[:, F*64:]).abs().max()) / ws:.3e})")
