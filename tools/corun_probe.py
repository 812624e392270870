"""How much does a bandwidth kernel slow down next to an MFMA GEMM running on another stream?
Stream B loops an NT GEMM (1x1 conv forward, level-3 shape); stream A times LayerNorm forward / backward alone and co-running."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from dcpt_amd import functional as DF
dev = torch.device('cuda:0')
B, C, H, W = 32, 512, 32, 32
x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
go = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
gx = torch.randn(B, 1024, H, W, device=dev).contiguous(memory_format=torch.channels_last); gw = torch.randn(512, 1024, 1, 1, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def gemm_loop(n):
    with torch.cuda.stream(sB), torch.no_grad():
        for _ in range(n): DF.conv_nobias(gx, gw)
def time_on_A(fn, n=40):
    with torch.cuda.stream(sA):
        fn(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(sA)
        for _ in range(n): fn()
        e1.record(sA)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def fwd():
    with torch.no_grad(): DF.layernorm2d(x, w, b, 1e-6)
y = DF.layernorm2d(x, w, b, 1e-6)
def bwd():
    x.grad = None; w.grad = None; b.grad = None
    y.backward(go, retain_graph=True)
for name, fn in (("ln fwd", fwd), ("ln fwd+bwd graph (bwd kernels)", bwd)):
    torch.cuda.synchronize()
    alone = time_on_A(fn)
    gemm_loop(400)          # ~400 x 260 us queued on stream B
    co = time_on_A(fn)
    torch.cuda.synchronize()
    print(f"{name:34s} alone {alone:8.1f} us   next to the GEMM stream {co:8.1f} us   x{co/alone:.2f}")
t0 = time.perf_counter(); gemm_loop(100); torch.cuda.synchronize(); print("GEMM alone %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
