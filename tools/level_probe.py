"""Time one NAFBlock forward / backward at every level of the bench configuration (B=32, 256^2, width 64).
    python tools/level_probe.py [bf16|x3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
BF = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
ES = 2 if BF else 4
if len(sys.argv) > 1 and sys.argv[1] == 'x3':
    DF.set_gemm_precision('bf16x3')
dev = torch.device('cuda:0')
def ev(): return torch.cuda.Event(enable_timing=True)
tot = 0.0
for lvl, (c, hw, nblk) in enumerate([(64, 256, 2), (128, 128, 2), (256, 64, 2), (512, 32, 29), (1024, 16, 1)]):
    blk = fill_module_(NAFBlock(c)).to(dev)
    x = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    go = torch.randn(32, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
    if BF:   # through the MODULE, as the network runs it: the per-block packed-weights cache on (round-4 verdict: the bare call re-packed per call)
        x, go = x.bfloat16(), go.bfloat16()
        blk.act_bf16 = True
    fwd = blk
    x.requires_grad_(True)
    for _ in range(3):
        y = fwd(x); y.backward(go)
    n = 10
    e0, e1, e2 = ev(), ev(), ev()
    tf = tb = 0.0
    ys = []
    e0.record()
    for _ in range(n):   # (no synchronisation inside the loops: back-to-back launches, as in a step)
        ys.append(fwd(x))
    e1.record()
    for y in ys:
        y.backward(go)
    e2.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / n; tb = e1.elapsed_time(e2) / n
    M = 32 * hw * hw
    gf = 36.0 * M * c * c / 1e9   # fwd+bwd GEMM flops of a block (6 c^2 MACs per pixel forward, x3)
    unit = M * c * ES / 1e6       # MB of one [M][C] tensor
    print(f"level {lvl}: C={c:4d} {hw}x{hw}  fwd {tf*1e3:7.0f} us  bwd {tb*1e3:7.0f} us  x{nblk:2d} blocks = {(tf+tb)*nblk:6.1f} ms   "
          f"GEMM {gf:6.1f} GF -> {gf/(tf+tb):6.1f} TF/s    [M][C] = {unit:6.1f} MB -> {(tf+tb)*1e-3*5.0e12/ (unit*1e6):5.1f} tensor passes at 5 TB/s")
    tot += (tf + tb) * nblk
print(f"blocks total {tot:.1f} ms")
