"""Times of the network-edge 3 x 3 convs (intro forward / backward, ending forward / backward) at the bench shapes, fp32 and bf16 features:
python tools/edge_times.py [B] [S]   (DCPT_EDGE_MFMA=0 for the VALU kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import _variant  # noqa: F401  (DCPT_TOOL_LIB=<variant library> selects a tools/build_variant.sh build)
from dcpt_amd import functional as DF
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for bf in (False, True):
    img = torch.rand((B, 3, S, S), device=dev)
    wi = (torch.randn((64, 3, 3, 3), device=dev) * 0.1).requires_grad_(True); bi = torch.zeros(64, device=dev, requires_grad=True)
    we = (torch.randn((3, 64, 3, 3), device=dev) * 0.1).requires_grad_(True); be = torch.zeros(3, device=dev, requires_grad=True)
    y = DF.conv3x3_in(img, wi, bi, out_bf16=bf)
    go = torch.rand_like(y)
    t_if = timeit(lambda: DF.conv3x3_in(img, wi, bi, out_bf16=bf))
    def ib():
        y = DF.conv3x3_in(img, wi, bi, out_bf16=bf); y.backward(go)
    t_ib = timeit(ib) - t_if
    f = y.detach().requires_grad_(True)
    t_ef = timeit(lambda: DF.conv3x3_out(f, we, be, img))
    go2 = torch.rand_like(img)
    def eb():
        o = DF.conv3x3_out(f, we, be, img); o.backward(go2)
    t_eb = timeit(eb) - t_ef
    print(f"{'bf16' if bf else 'fp32'} B={B} {S}x{S}: intro fwd (s2b) {t_if:7.1f} us  intro bwd (wgrad) {t_ib:7.1f} us  ending fwd (b2s) {t_ef:7.1f} us  ending bwd (s2b + wgrad) {t_eb:7.1f} us")
