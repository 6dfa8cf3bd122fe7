"""rp_linear_fwd_pieces (pre-split bf16 operands, LDS-DMA staging; csrc/gemm_pieces.hip) against rp_linear_fwd (fp32 operands
split in the kernel) at the wide-MLP shapes.  np = 2 must be bit-identical to the bf16x3 mode.  Run on the GPU box:
    python profiles/microbench/probes/probe_pieces.py [quick]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from rec_pangu_amd import hip  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


shapes = [(65536, 1024, 1677), (65536, 512, 1024), (65536, 256, 512), (65536, 1677, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = [(1000, 300, 100), (4096, 1024, 1677)]
for M, N, K in shapes:
    ldx = (K + 3) // 4 * 4
    x = torch.randn(M, ldx, device=dev)[:, :K]
    w = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    flops = 2.0 * M * N * K
    for mode, np_ in (("bf16x3", 2), ("bf16", 1)):
        hip.set_matmul_precision(mode)
        ref = hip.linear_fwd(x, w, bias, act=hip.ACT_RELU)
        t_ref = timed(lambda: hip.linear_fwd(x, w, bias, act=hip.ACT_RELU))
        xp = hip.pieces_pack(x, np_)
        wp = hip.pieces_pack(w, np_)
        out = hip.linear_fwd_pieces(xp, wp, bias, K, np_, act=hip.ACT_RELU)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        err = float((out - ref).abs().max())
        t_new = timed(lambda: hip.linear_fwd_pieces(xp, wp, bias, K, np_, act=hip.ACT_RELU))
        t_px = timed(lambda: hip.pieces_pack(x, np_, out=xp))
        t_pw = timed(lambda: hip.pieces_pack(w, np_, out=wp))
        print(f"[{M}x{N}x{K}] {mode:7s} fp32-operand kernel {t_ref:.4f} ms = {flops / t_ref / 1e9:7.1f} TFLOP/s | pieces {t_new:.4f} ms = "
              f"{flops / t_new / 1e9:7.1f} TFLOP/s | bit-identical {same} (max diff {err:.3g}) | pack x {t_px:.4f} ms, pack w {t_pw:.4f} ms",
              flush=True)
hip.set_matmul_precision("auto")
