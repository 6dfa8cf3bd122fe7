"""Scratch: run one GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py fwd|wgrad N K lda [mode]"""
import sys, torch
sys.path.insert(0, ".")
from rec_pangu_amd import hip
kind, N, K, lda = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hip.set_matmul_precision(sys.argv[5] if len(sys.argv) > 5 else "bf16x6")
M = 65536
a = torch.randn(M, lda, device="cuda")
if kind == "fwd":
    w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(3): hip.linear_fwd(a, w, None, 0, K=K, out=out)
else:
    dy = torch.randn(M, N, device="cuda")
    for _ in range(3): hip.linear_wgrad(dy, a, K)
torch.cuda.synchronize()
