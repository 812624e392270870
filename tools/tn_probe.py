"""TN (wgrad) GEMM speed by loader at the level-3 shape, through the public conv ops' backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from dcpt_amd import functional as DF, _lib
import ctypes
lib = _lib.load(); lib.dcpt_set_side_stream(0)
dev = torch.device('cuda:0')
B, H, W = 32, 32, 32
for Ci, Co in [(512, 1024), (1024, 512), (512, 512)]:
    x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(False)
    w = torch.randn(Co, Ci, 1, 1, device=dev, requires_grad=True)
    y = DF.conv_nobias(x, w); gy = torch.randn_like(y)
    lib.dcpt_prof_enable(1)
    for _ in range(20):
        w.grad = None
        y = DF.conv_nobias(x, w); y.backward(gy)
    torch.cuda.synchronize()
    buf = (ctypes.c_double * (8 * 64))(); n = lib.dcpt_prof_read(buf, 64); lib.dcpt_prof_enable(0)
    for i in range(n):
        cls, M, N, K, cnt, ms, fl, by = (buf[i * 8 + j] for j in range(8))
        print(f"Ci={Ci} Co={Co} class {int(cls)} M={int(M)} N={int(N)} K={int(K)} launches {int(cnt)}  {ms/cnt*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s")
