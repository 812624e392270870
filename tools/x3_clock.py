"""Shader clock held during the main loop of the bf16x3 NT kernel (variant libraries built with -DX3_ABL_CLOCK write block 0's
s_memtime / s_memrealtime deltas into C[0][0..1]).   DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_x3_<v>.so python tools/x3_clock.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")
torch.manual_seed(0)
DF.set_gemm_precision("bf16x3")
for (B, H, W, Ci, Co) in [(32, 32, 32, 1024, 512), (32, 32, 32, 512, 1024)]:
    x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5
    with torch.no_grad():
        for _ in range(20): y = DF.conv_nobias(x, w)   # warm: the clock settles under load
        torch.cuda.synchronize()
        v = y.permute(0, 2, 3, 1).reshape(-1, Co)[0, :2].tolist()
    print(f"{os.environ.get('DCPT_TOOL_LIB', 'product').split('_x3_')[-1]:14s} K={Ci} N={Co}: loop {v[0]:.0f} shader cycles in {v[1] * 0.01:.1f} us -> {v[0] / (v[1] * 10.0) :.3f} GHz")
