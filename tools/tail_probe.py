"""Tail cost of the level-3 NT grids: time the plain 1x1-conv GEMM (K = N = 512) at row counts whose 128 x 64 tile grids are a whole
number of rounds of the 768 resident slots (3 per CU) and at the bench's M = 32768 (2048 tiles = 2.67 rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")
C = 512
w = torch.randn(C, C, 1, 1, device=dev) * 0.02
for M in (24576, 32768, 36864, 49152, 65536, 73728):
    H = M // 32 // 32
    x = torch.randn(32, C, H, 32, device=dev).contiguous(memory_format=torch.channels_last)
    assert x.shape[0] * x.shape[2] * x.shape[3] == M
    for _ in range(5): y = DF.conv_nobias(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): y = DF.conv_nobias(x, w)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    tiles = (M // 128) * (C // 64)
    print(f"M={M:6d} tiles={tiles:5d} = {tiles/768:5.2f} rounds of 768: {us:7.1f} us  {2.0*M*C*C/us/1e6:6.1f} TF/s")
