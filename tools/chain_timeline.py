"""Phase timeline of the chain kernel (variant built with -DCHAIN_TIMELINE): shader-clock stamps of waves 0 / 4 of every block.
    DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_chain_tl.so python tools/chain_timeline.py <level> <train|infer>"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import _lib, functional as DF
lvl = int(sys.argv[1]); c, hw = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)][lvl]
train = sys.argv[2] == "train"
dev = torch.device('cuda:0')
blk = fill_module_(NAFBlock(c)).to(dev)
x = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).bfloat16()
packed = DF.PackedWeightsBf16(); P = blk.fused_params()
for _ in range(4):
    if train:
        y = DF.nafblock_bf16(x.requires_grad_(True), P, packed)
    else:
        with torch.no_grad():
            y = DF.nafblock_bf16(x, P, packed)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros(512 * 2 * 24, dtype=np.uint64)
assert lib.dcpt_chain_timeline_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
t = buf.reshape(512, 2, 24)[:256].astype(np.int64)
names = ["start", "tab barrier", "rows in LDS / conv3 part done (y stored)", "LN done", "B1 passed", "pass0 loop", "pass0 epi", "pass1 loop", "pass1 epi", "B2 passed", "g written", "B3 passed",
         "conv5 loop", "conv5 epi issued", "stores drained", "-", "c3: t2 rows in LDS", "c3: barrier", "c3: conv3 loop", "c3: inp requested + barrier", "c3: y -> LDS", "c3: barrier", "-", "-"]
order = [0, 1, 16, 17, 18, 19, 20, 21, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
t0 = t[:, :, 0:1]
d = t - t0
print("phase end, cycles since the wave's start: median over blocks [min .. max]  (wave 0 | wave 4);   delta to previous (median)")
prev = np.zeros(2)
for i in order:
    n = names[i]
    a, b = d[:, 0, i], d[:, 1, i]
    if a.max() <= 0 and i >= 16:
        continue
    ma, mb = np.median(a), np.median(b)
    print(f"{n:22s} {ma:9.0f} [{a.min():7d} .. {a.max():7d}] | {mb:9.0f} [{b.min():7d} .. {b.max():7d}]    +{ma - prev[0]:8.0f}  +{mb - prev[1]:8.0f}")
    prev = np.array([ma, mb])
print("block start spread (cycles):", int(t[:, 0, 0].max() - t[:, 0, 0].min()))
