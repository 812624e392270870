"""A/B of the fused depthwise backward: register kernel (DCPT_DW_RING_BWD=0) vs ring kernel, through one NAFBlock backward.
    DCPT_DW_RING_BWD=0 python tools/dwrb_check.py save /tmp/ref.pt ; python tools/dwrb_check.py cmp /tmp/ref.pt"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")
SHAPES = [(1, 8, 4, 6), (2, 16, 16, 16), (2, 16, 13, 11), (1, 32, 64, 40), (3, 24, 9, 70), (2, 64, 40, 130), (1, 16, 50, 33),
          (2, 128, 32, 32), (1, 256, 20, 64), (2, 64, 70, 16), (1, 8, 96, 200)]
res = {}
for bf in (False, True):
    for (B, c, H, W) in SHAPES:
        if bf and c % 8:
            continue
        torch.manual_seed(1)
        blk = fill_module_(NAFBlock(c)).to(dev)
        x = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        go = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        ps = blk.fused_params()
        if bf:
            x, go = x.bfloat16(), go.bfloat16()
        x.requires_grad_(True)
        y = DF.nafblock_bf16(x, ps) if bf else blk(x)
        y.backward(go)
        torch.cuda.synchronize()
        res[(bf, B, c, H, W)] = dict(dx=x.grad.float().cpu(), dw2=blk.conv2.weight.grad.cpu(), db2=blk.conv2.bias.grad.cpu(),
                                     dw1=blk.conv1.weight.grad.cpu(), dn1=blk.norm1.weight.grad.cpu())
if sys.argv[1] == "save":
    torch.save(res, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    worst = 0.0
    for k in res:
        line = []
        for n in res[k]:
            a, b = res[k][n], ref[k][n]
            e = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
            tol = 2e-2 if k[0] else 2e-5
            worst = max(worst, e / tol)
            line.append(f"{n} {e:.2e}")
        print(k, "  ".join(line))
    print("WORST (err / tol):", worst, "OK" if worst < 1 else "FAIL")
