"""Library fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) on the level-3 shapes of the training step, next to this repo's
MFMA GEMMs on the same shapes (1x1 conv forward = NT, weight gradient = TN).  Context for DESIGN.md, not part of the product."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from dcpt_amd import functional as DF, _lib
dev = torch.device("cuda:0")
lib = _lib.load(); lib.dcpt_set_side_stream(0)
torch.backends.cuda.matmul.allow_tf32 = False


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


import ctypes
import bench


def kernel_tflops(fn):
    """kernel-only rate of this repo's GEMM launches inside fn() (library HIP-event profiler, per launch class)"""
    fn(); torch.cuda.synchronize()
    lib.dcpt_prof_enable(1)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_double * (8 * 64))(); n = lib.dcpt_prof_read(buf, 64); lib.dcpt_prof_enable(0)
    out = {}
    for i in range(n):
        cls, M, N, K, cnt, ms, fl, by = (buf[i * 8 + j] for j in range(8))
        out[bench.prof_class_name(int(cls)).split("<")[0]] = fl / ms / 1e9
    return out


for M, N, K in ((32768, 512, 1024), (32768, 1024, 512), (32768, 512, 512), (131072, 256, 256), (2097152, 128, 64)):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); g = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    t_lib_nt = timed(lambda: torch.mm(a, w.t()))
    t_lib_tn = timed(lambda: torch.mm(g.t(), a))
    hw = 32 if M % 1024 == 0 and M <= 262144 else 256
    B = M // (hw * hw)
    x = a.view(B, hw, hw, K).permute(0, 3, 1, 2)          # NHWC memory, NCHW view
    wr = w.view(N, K, 1, 1).detach().requires_grad_(True)
    go = g.view(B, hw, hw, N).permute(0, 3, 1, 2)
    y = DF.conv_nobias(x, wr)
    mine = kernel_tflops(lambda: torch.autograd.grad(DF.conv_nobias(x, wr), wr, go))   # forward NT, dgrad NT, wgrad TN
    print(f"M={M:7d} N={N:4d} K={K:4d} | library (torch.mm) NT {fl/t_lib_nt/1e12:6.1f}  TN {fl/t_lib_tn/1e12:6.1f} TF/s | this repo, kernel only: "
          + "  ".join(f"{k} {v:6.1f}" for k, v in sorted(mine.items())) + " TF/s")
