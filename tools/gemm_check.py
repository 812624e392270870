"""Spot-check the NT GEMM (1x1 conv, no bias) against torch on awkward shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from dcpt_amd import functional as DF
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, H, W, Ci, Co) in [(1, 16, 16, 64, 128), (1, 16, 16, 32, 32), (1, 16, 16, 16, 32), (1, 16, 16, 32, 16), (1, 8, 8, 32, 32), (1, 16, 16, 48, 32), (2, 12, 20, 16, 16), (1, 16, 16, 128, 128), (1,16,16,36,32)]:
    x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(Co, Ci, 1, 1, device=dev)
    with torch.no_grad():
        y = DF.conv_nobias(x, w); r = torch.nn.functional.conv2d(x.double(), w.double()).float()
    err = (y - r).abs().max().item() / r.abs().max().item()
    bad = (~torch.isfinite(y)).sum().item()
    print(f"M={B*H*W:5d} K={Ci:4d} N={Co:4d}  rel err {err:.2e}  nonfinite {bad}")
