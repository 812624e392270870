"""Forward of one NAFBlock at a level of the bench configuration in bf16 storage, 10 times (run under rocprofv3 --kernel-trace; the chain
kernel of the wide levels is what this is for).   python tools/chain_trace.py <level> <train|infer> [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
lvl = int(sys.argv[1]); c, hw = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)][lvl]
train = sys.argv[2] == "train"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device('cuda:0')
blk = fill_module_(NAFBlock(c)).to(dev)
x = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).bfloat16()
packed = DF.PackedWeightsBf16()
P = blk.fused_params()
for _ in range(10):
    if train:
        y = DF.nafblock_bf16(x.requires_grad_(True), P, packed)
    else:
        with torch.no_grad():
            y = DF.nafblock_bf16(x, P, packed)
torch.cuda.synchronize()
