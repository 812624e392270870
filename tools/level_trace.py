"""One NAFBlock forward + backward at one level of the bench configuration, 5 times (run under rocprofv3 --kernel-trace).
    python tools/level_trace.py <level 0..4> [bf16|x3]   (x3: fp32 storage, GEMMs in the bf16x3 split-operand mode; SIDE=1 in the environment: weight gradients on the side stream)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import _lib, functional as DF
_lib.load().dcpt_set_side_stream(int(os.environ.get("SIDE", "0")))
lvl = int(sys.argv[1]); c, hw = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)][lvl]
bf = len(sys.argv) > 2 and sys.argv[2] == "bf16"
dev = torch.device('cuda:0')
blk = fill_module_(NAFBlock(c)).to(dev)
B = int(os.environ.get('B', '32'))
x = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
go = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
if bf:
    x, go = x.bfloat16(), go.bfloat16()
if len(sys.argv) > 2 and sys.argv[2] == "x3":
    DF.set_gemm_precision("bf16x3")
x.requires_grad_(True)
for _ in range(5):
    y = DF.nafblock_bf16(x, blk.fused_params()) if bf else blk(x)
    y.backward(go)
torch.cuda.synchronize()
