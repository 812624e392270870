"""Is the fp32-MFMA GEMM at the power-limited clock?  Same launch on random vs zero operands (MI355X_MICROARCH.md, DVFS)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
dev = torch.device('cuda:0')
B, H, W, Ci, Co = 32, 32, 32, 1024, 512
def run(x, w, n=30):
    for _ in range(3): DF.conv_nobias(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): DF.conv_nobias(x, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, 2.0 * B * H * W * Ci * Co / ms / 1e9
for name, mk in [("random N(0,1)", lambda s: torch.randn(s, device=dev)), ("zeros", lambda s: torch.zeros(s, device=dev)),
                 ("ones", lambda s: torch.ones(s, device=dev)), ("random again", lambda s: torch.randn(s, device=dev))]:
    x = mk((B, Ci, H, W)).contiguous(memory_format=torch.channels_last); w = mk((Co, Ci, 1, 1))
    with torch.no_grad():
        ms, tf = run(x, w)
    print(f"{name:16s} {ms*1e3:8.1f} us  {tf:7.1f} TF/s", flush=True)
