"""One GEMM shape a few times (for rocprofv3 --pmc): gemm_one.py M N K lda mode [reps]"""
import sys
import torch
sys.path.insert(0, '.')
from rec_pangu_amd import hip
M, N, K, lda, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
hip.lib()
hip.set_matmul_precision(mode)
a = torch.randn(M, lda, device='cuda')
w = torch.randn(N, (K + 3) // 4 * 4, device='cuda')[:, :K]
b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda')
for _ in range(reps):
    hip.linear_fwd(a, w, b, hip.ACT_RELU, K=K, out=out)
torch.cuda.synchronize()
