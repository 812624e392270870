"""NAFBlock forward, ring depthwise kernel vs register kernel (DCPT_DW_RING=0 in a second process is the reference): prints max diff
of the block output for a few shapes.  usage: python tools/dwr_check.py [save|cmp] file"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")
outs = {}
for bf in (False, True):
    for (B, c, H, W) in [(2, 16, 16, 16), (3, 24, 5, 7), (2, 8, 40, 70), (2, 64, 136, 136), (1, 512, 68, 68), (1, 1024, 34, 33), (4, 128, 272, 270), (1, 64, 544, 544), (9, 64, 528, 544), (3, 256, 136, 132)]:
        torch.manual_seed(0)
        blk = fill_module_(NAFBlock(c)).to(dev)
        x = torch.randn(B, c, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            y = DF.nafblock_bf16(x.bfloat16(), blk.fused_params()).float() if bf else blk(x)
        outs[f"{'bf16' if bf else 'fp32'}_{B}_{c}_{H}_{W}"] = y.cpu()
if sys.argv[1] == "save":
    torch.save(outs, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    for k in outs:
        d = float((outs[k] - ref[k]).abs().max() / ref[k].abs().max())
        print(f"{k:24s} rel diff {d:.3e}")
