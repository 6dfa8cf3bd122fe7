"""One weight-gradient shape, timed with HIP events: wgrad_one.py M N K ldx mode [reps]   (dW[N,K] = dY[M,N]^T . X[M,K])"""
import sys
import torch
sys.path.insert(0, '.')
from rec_pangu_amd import hip
M, N, K, ldx, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
hip.lib()
hip.set_matmul_precision(mode)
x = torch.randn(M, ldx, device='cuda')
dy = torch.randn(M, N, device='cuda')
big = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
for _ in range(3):
    hip.linear_wgrad(dy, x, K, want_bias=True)
ts = []
for _ in range(reps):
    big.zero_()  # evict x from the MALL, as a step's other kernels would
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.linear_wgrad(dy, x, K, want_bias=True)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
ms = ts[len(ts) // 2]
print(f"wgrad {M}x{N}x{K} {mode}: {ms:.4f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s  {(M * (K + N) * 4) / ms / 1e6:.0f} GB/s")
