import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
import torch.nn.functional as F
dev = torch.device('cuda:0'); torch.manual_seed(0)
B, C, H, W = 1, 64, 16, 16
x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(2 * C, C, 1, 1, device=dev)
skip = torch.randn(B, C // 2, 2 * H, 2 * W, device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    y = DF.up_ps(x, w, skip)
    r = F.pixel_shuffle(F.conv2d(x, w), 2) + skip
    r0 = F.pixel_shuffle(F.conv2d(x, w), 2)
d = (y - r).abs()
print("max err", d.max().item(), " err vs no-skip", (y - r0).abs().max().item(), " y==skip?", (y - skip).abs().max().item())
bad = (d > 1e-3).nonzero()
print("bad count", len(bad), "of", d.numel()); print(bad[:10].tolist())
# residual epilogue: conv with residual? use nafblock zero gain
