"""one shape of rp_linear_fwd_pieces, timing only (RP_PIECES_DEBUG=1: no loads in the loop, 2: no matrix work)
    python profiles/microbench/probes/probe_pieces_one.py [np] [M N K]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from rec_pangu_amd import hip  # noqa: E402

dev = torch.device("cuda")
np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M, N, K = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (65536, 1024, 1677)
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
xp, wp = hip.pieces_pack(x, np_), hip.pieces_pack(w, np_)
out = torch.empty(M, N, device=dev)
for _ in range(3):
    hip.linear_fwd_pieces(xp, wp, None, K, np_, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    hip.linear_fwd_pieces(xp, wp, None, K, np_, out=out)
b.record()
torch.cuda.synchronize()
t = a.elapsed_time(b) / 10
print(f"np={np_} [{M}x{N}x{K}] dbg={os.environ.get('RP_PIECES_DEBUG', '0')}: {t:.4f} ms = {2.0 * M * N * K / t / 1e9:.1f} TFLOP/s", flush=True)
