"""How much of rp_embed_grad_gemm's time is the random per-pair gather of dH / sum rows?  Same launch at Criteo shape with
(a) the real (row-sorted) positions, (b) positions replaced by a sequential walk inside each field (gathers become
coalesced streams; the sorted keys, i.e. run structure and gradient-row writes, stay as they are — results are garbage,
timing is the point), (c) without the FM term (no sum rows, no arena rows).  Run on the GPU box:
    python profiles/microbench/probes/probe_grad_gemm.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench  # noqa: E402
from rec_pangu_amd import hip  # noqa: E402

dev = torch.device("cuda")
enc = bench.criteo_enc_dict(1)
B, D = 65536, 64
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
dh = torch.randn(B, 64, device=dev)
wt = torch.randn(F * 64 + 64, 64, device=dev)
gfm = torch.randn(B, device=dev)
ssum = torch.randn(B, D, device=dev)
i = torch.arange(F * B, device=dev, dtype=torch.int64)
sp_seq = ((i // B) * B + (i % B)).to(torch.int32)  # == arange: position i sits in field i // B, sample i % B
assert torch.equal(sp_seq.long(), i)


def run(name, spx, fm=True, n=20):
    for _ in range(3):
        hip.embed_grad_gemm(sk, spx, B, D, dh, wt, None, gfm if fm else None, ssum if fm else None, arena, G, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        hip.embed_grad_gemm(sk, spx, B, D, dh, wt, None, gfm if fm else None, ssum if fm else None, arena, G, False)
    b.record()
    torch.cuda.synchronize()
    print(f"{name:55s} {a.elapsed_time(b) / n:.4f} ms")


run("real positions (row-sorted), FM term", sp)
run("sequential positions inside each field, FM term", sp_seq)
run("real positions, no FM term (no sum / arena rows)", sp, fm=False)
run("sequential positions, no FM term", sp_seq, fm=False)
# the same with only the 18 non-tiny fields' pairs (what the kernel would see if tiny tables went elsewhere)
tiny = [f for f in range(F) if rows[f] <= 257]
keep = torch.ones(F * B, dtype=torch.bool, device=dev)
for f in tiny:
    keep[f * B:(f + 1) * B] = False
sk18, sp18 = sk[keep].contiguous(), sp[keep].contiguous()


def run18(name, n=20):
    for _ in range(3):
        hip.embed_grad_gemm(sk18, sp18, B, D, dh, wt, None, gfm, ssum, arena, G, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        hip.embed_grad_gemm(sk18, sp18, B, D, dh, wt, None, gfm, ssum, arena, G, False)
    b.record()
    torch.cuda.synchronize()
    print(f"{name:55s} {a.elapsed_time(b) / n:.4f} ms   ({len(tiny)} tiny fields = {len(tiny) * B} pairs left out)")


run18("real positions, FM term, non-tiny fields only")

# round 4: the tiny tables on the sample-major one-hot path, the row-sorted kernel over the other fields
tiny_t = [(f, int(base[f]), rows[f]) for f in tiny]
skip = sum(1 << f for f in tiny)


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:55s} {a.elapsed_time(b) / n:.4f} ms")


timeit("rp_embed_grad_tiny alone (8 tables, 2 launches)", lambda: hip.embed_grad_tiny(keys, B, tiny_t, dh, wt, gfm, ssum, arena, G, False))
timeit("rp_embed_grad_gemm(skip_fields) alone", lambda: hip.embed_grad_gemm(sk, sp, B, D, dh, wt, None, gfm, ssum, arena, G, False, skip_fields=skip))
timeit("both, back to back", lambda: (hip.embed_grad_tiny(keys, B, tiny_t, dh, wt, gfm, ssum, arena, G, False),
                                      hip.embed_grad_gemm(sk, sp, B, D, dh, wt, None, gfm, ssum, arena, G, False, skip_fields=skip)))
