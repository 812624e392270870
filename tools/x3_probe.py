"""bf16x3 split-operand NT GEMM: accuracy against fp64 and timing against the fp32-MFMA kernel (level-3 shapes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")
torch.manual_seed(0)
def ev(): return torch.cuda.Event(enable_timing=True)
for (B, H, W, Ci, Co) in [(32, 32, 32, 512, 1024), (32, 32, 32, 1024, 512), (32, 32, 32, 512, 512), (32, 64, 64, 256, 512), (25, 31, 32, 512, 512)]:
    x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5
    M = B * H * W
    ref = (x.permute(0, 2, 3, 1).reshape(M, Ci).double() @ w.reshape(Co, Ci).double().t())
    out = {}
    for mode in ("fp32", "bf16x3"):
        DF.set_gemm_precision(mode)
        with torch.no_grad():
            y = DF.conv_nobias(x, w)
            for _ in range(3): DF.conv_nobias(x, w)
            e0, e1 = ev(), ev(); e0.record()
            for _ in range(10): DF.conv_nobias(x, w)
            e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        yy = y.permute(0, 2, 3, 1).reshape(M, Co).double()
        err = (yy - ref).abs().max().item() / ref.abs().max().item()
        rms = ((yy - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
        out[mode] = (t, err, rms)
    gf = 2.0 * M * Ci * Co / 1e9
    print(f"M={M} K={Ci} N={Co}: fp32 {out['fp32'][0]:7.1f} us ({gf/out['fp32'][0]*1e-3:6.1f} TF/s) max-rel err {out['fp32'][1]:.2e} rms {out['fp32'][2]:.2e} | "
          f"bf16x3 {out['bf16x3'][0]:7.1f} us ({gf/out['bf16x3'][0]*1e-3:6.1f} TF/s) err {out['bf16x3'][1]:.2e} rms {out['bf16x3'][2]:.2e}")
DF.set_gemm_precision("fp32")
