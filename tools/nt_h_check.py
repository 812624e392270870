"""Checksums of the bf16 block / conv + LayerNorm / bottleneck results on shapes that reach the 256-row NT kernels -- run once per kernel routing
(DCPT_TOOL_LIB variant build + DCPT_NT_H = 0 | 1 | 2) and diff the outputs: the 256 x 128 two-blocks-per-CU kernel accumulates in the same order
as the 256 x 256 kernel, so every line must be identical."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tools"))
import _variant  # noqa
import torch
from kernel_trace import kernel_trace
from dcpt_amd import functional as DF
import test_gpu_bf16 as T
dev = torch.device("cuda:0")
def h(*ts):
    m = hashlib.sha1()
    for t in ts:
        m.update(t.detach().float().cpu().numpy().tobytes())
    return m.hexdigest()[:12]
g = torch.Generator(device=dev).manual_seed(5)
for shape in [(24, 512, 32, 32), (25, 512, 31, 32), (13, 512, 48, 40), (48, 256, 32, 32), (32, 128, 128, 128), (6, 1024, 16, 16)]:
    B, c, H, W = shape
    P = T._params(c, "h.")
    Pd = {k: P[v].to(dev).requires_grad_(True) for k, v in T.FUSED.items()}
    x = (torch.rand(shape, generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gw = (torch.rand(shape, generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    with kernel_trace() as tr:
        y = DF.nafblock_bf16(x, Pd); y.backward(gw); torch.cuda.synchronize()
    print("block", shape, h(y, x.grad, *[Pd[k].grad for k in T.FUSED]), {k: v for k, v in tr.counts.items() if k.startswith("nt_bf16")})
for (B, Cin, Cout, H, W, ks) in [(2, 128, 128, 225, 223, 3), (2, 256, 128, 224, 224, 1), (3, 256, 256, 128, 128, 3), (2, 256, 512, 64, 64, 3), (4, 512, 512, 32, 32, 3), (1, 64, 128, 256, 256, 1)]:
    x = (torch.rand((B, Cin, H, W), generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.rand((Cout, Cin, ks, ks), generator=g, device=dev) - 0.5).requires_grad_(True)
    lw = torch.rand(Cout, generator=g, device=dev).requires_grad_(True); lb = torch.rand(Cout, generator=g, device=dev).requires_grad_(True)
    gw = (torch.rand((B, Cout, H, W), generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    with kernel_trace() as tr:
        y = DF.conv_ln_bf16(x, w, lw, lb, None, True); y.backward(gw); torch.cuda.synchronize()
    print("convln", (B, Cin, Cout, H, W, ks), h(y, x.grad, w.grad, lw.grad, lb.grad), {k: v for k, v in tr.counts.items() if k.startswith("nt_bf16")})
for (B, C, H, W) in [(2, 64, 225, 223), (1, 128, 224, 224), (4, 256, 64, 64)]:
    x = (torch.rand((B, C, H, W), generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ws = []
    for (co, ci, k) in ((2 * C, C, 1), (2 * C, 2 * C, 3), (C, 2 * C, 1)):
        ws += [(torch.rand((co, ci, k, k), generator=g, device=dev) - 0.5).requires_grad_(True), torch.rand(co, generator=g, device=dev).requires_grad_(True), torch.rand(co, generator=g, device=dev).requires_grad_(True)]
    gw = (torch.rand((B, C, H, W), generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    with kernel_trace() as tr:
        y = DF.bottleneck(x, *ws); y.backward(gw); torch.cuda.synchronize()
    print("bneck", (B, C, H, W), h(y, x.grad, *[t.grad for t in ws]), {k: v for k, v in tr.counts.items() if k.startswith("nt_bf16")})
