import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
from dcpt_amd._lib import PARAM_FIELDS
dev = torch.device("cuda:0")
class Ctx:
    def save_for_backward(self, *a): self.saved = a
res = {}
for (B, c, H, W) in [(1, 8, 4, 6), (2, 16, 16, 16)]:
    torch.manual_seed(0)
    blk = fill_module_(NAFBlock(c)).to(dev)
    x = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    ctx = Ctx()
    ps = blk.fused_params()
    with torch.no_grad():
        out = DF._NAFBlockFn.forward(ctx, x, *[ps[k] for k in PARAM_FIELDS])
    inp, t1, t2, y, v, stats, pooled, s, xn = ctx.saved[:9]
    res[(B, c, H, W)] = dict(t1=t1.cpu(), t2=t2.cpu(), pooled=pooled.cpu())
if sys.argv[1] == "save":
    torch.save(res, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    for k in res:
        for n in ("t1", "t2", "pooled"):
            a, b = res[k][n], ref[k][n]
            print(k, n, float((a - b).abs().max()), float(b.abs().max()))
        a, b = res[k]["t2"], ref[k]["t2"]
        bad = (a - b).abs() > 1e-5
        idx = bad.nonzero()
        print(" bad count", int(bad.sum()), "of", bad.numel(), " first bad idx (n,c,h,w):", idx[:12].tolist())
