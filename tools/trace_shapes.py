"""Which kernel families does the bf16 NAFBlock / the head / the edge convs take at the parity suite's shapes?  (discovery run for the
expectations asserted in tests/test_gpu_bf16.py and tests/test_gpu_dispatch.py)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from kernel_trace import kernel_trace
from dcpt_amd import functional as DF
import test_gpu_bf16 as T

dev = torch.device("cuda:0")
shapes = [(2, 64, 32, 32), (3, 64, 5, 7), (5, 64, 48, 40), (2, 16, 6, 10), (1, 128, 16, 16), (2, 512, 8, 16), (1, 1024, 8, 8), (1, 16, 11, 70),
          (24, 512, 32, 32), (25, 512, 31, 32), (6, 256, 64, 64), (13, 512, 48, 40), (200, 512, 12, 12), (48, 256, 32, 32), (4, 128, 64, 64), (32, 128, 128, 128)]
for shape in shapes:
    B, c, H, W = shape
    P = T._params(c, "tr.")
    Pd = {k: P[v].to(dev).requires_grad_(True) for k, v in T.FUSED.items()}
    x = torch.randn(shape, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with kernel_trace() as tr:
        y = DF.nafblock_bf16(x, Pd)
        y.backward(torch.randn_like(y))
        torch.cuda.synchronize()
    print(shape, dict(sorted(tr.counts.items())))
