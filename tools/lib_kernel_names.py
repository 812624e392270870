"""Which library kernels torch.mm picks for the level-3 GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
dev = torch.device("cuda:0")
for M, N, K in ((32768, 512, 1024), (32768, 1024, 512), (32768, 512, 512)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); g = torch.randn(M, N, device=dev)
    for _ in range(3):
        torch.mm(a, w.t()); torch.mm(g.t(), a)
torch.cuda.synchronize()
