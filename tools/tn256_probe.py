"""Weight-gradient GEMM in isolation (dcpt_conv1x1_wgrad_bf16): time and TF/s of the level-2/3/4 shapes of NAFNet-64 at B = 32, 256^2,
kernel + finisher together (what replaces the 128-wide kernel + its reducer).   python tools/tn256_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
import torch
from dcpt_amd import functional as DF

dev = torch.device("cuda:0")
SHAPES = [(32768, 1024, 512), (32768, 512, 512), (131072, 512, 256), (131072, 256, 256), (8192, 2048, 1024), (65536, 1024, 512), (16384, 1024, 512)]
if os.environ.get("SHAPE"):   # one shape only (PMC passes: tools/exp_r5e.sh)
    SHAPES = [SHAPES[int(os.environ["SHAPE"])]]
for M, N, K in SHAPES:
    R = 6   # operand sets in rotation: ~600 MB per round, beyond the 256 MB Infinity Cache
    dys = [torch.randn((M, N), device=dev).bfloat16() for _ in range(R)]
    xs = [torch.randn((M, K), device=dev).bfloat16() for _ in range(R)]
    for i in range(3):
        DF.conv1x1_wgrad_bf16(dys[i], xs[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for i in range(n):
        DF.conv1x1_wgrad_bf16(dys[i % R], xs[i % R])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s (kernel + finisher)")
