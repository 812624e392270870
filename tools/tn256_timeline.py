"""Timeline of the grouped bf16 weight-gradient launch of one level-3 NAFBlock backward (diagnostic build:
tools/build_variant.sh tl256 "gemm_tn_bf16_256.hip nafblock_bf16.hip" -DTN256_TIMELINE; DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_tl256.so):
when do the sibling tiles of one pixel range (same problem, same range: they share operand columns through their XCD's L2) pass the
quarters of their k-loop -- in lockstep or apart?    python tools/tn256_timeline.py [level]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import _lib, functional as DF
lib = _lib.load(); lib.dcpt_set_side_stream(0)
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
c, hw = [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)][lvl]
dev = torch.device('cuda:0'); B = int(os.environ.get('B', '32'))
blk = fill_module_(NAFBlock(c)).to(dev)
x = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).bfloat16().requires_grad_(True)
go = torch.randn(B, c, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).bfloat16()
for _ in range(4):
    y = DF.nafblock_bf16(x, blk.fused_params()); y.backward(go)
torch.cuda.synchronize()
rd = lib.dcpt_timeline_read_tn256; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((1024, 10), dtype=np.uint64); assert rd(buf.ctypes.data, 1024) == 0
rows = buf[buf[:, 9] == 1]
t = rows[:, :7].astype(np.int64); t0 = t[:, 0].min(); t = (t - t0) * 10e-3   # us
info = rows[:, 8]; xcc = (info >> 48) & 15; prob = (info >> 32) & 0xffff; split = (info >> 16) & 0xffff; tile = info & 0xffff
print(f"{len(rows)} running blocks; launch span {t[:, 6].max():.1f} us; columns: start, prologue done, 1/4, 1/2, 3/4 of the first segment's k-loop, loop end, block end [us]")
names = ["conv5", "conv4", "conv3", "conv1"]
print("XCC ids seen per (hardware block id & 7):", {int(b): sorted(set(int(v) for v in xcc[np.arange(len(buf))[buf[:, 9] == 1] % 8 == b])) for b in range(8)})
worst = []
for p in sorted(set(prob)):
    for s in sorted(set(split[prob == p])):
        m = (prob == p) & (split == s)
        xs = sorted(set(int(v) for v in xcc[m]))
        sp = t[m][:, 1:6].max(axis=0) - t[m][:, 1:6].min(axis=0)   # spread of the siblings at each stamp
        worst.append((sp.max(), p, s))
        if s < 3:
            print(f"{names[int(p)]} range {int(s)}: {int(m.sum())} tiles on XCC {xs}; sibling spread at the stamps {np.round(sp, 1)} us; start {np.round(t[m][:, 0], 1)}")
# k-loop duration by tile index within the pixel range (is it the SAME tiles that are slow everywhere?  tile = tile_n * tiles_k + tile_k; tile_k == 0 tiles
# also sum X's columns)
for p in sorted(set(prob)):
    m = prob == p
    nt = int(tile[m].max()) + 1
    per = [np.median((t[:, 5] - t[:, 1])[m & (tile == k)]) for k in range(nt)]
    print(f"{names[int(p)]}: median k-loop (first segment) by tile index: {np.round(per, 1)} us")
    s0 = m & (split == 0)
    o = np.argsort(tile[s0])
    print(f"   range 0 per tile: loop {np.round((t[:, 5] - t[:, 1])[s0][o], 1)}  hw block ids {np.arange(len(buf))[buf[:, 9] == 1][s0][o]}")
sp_all = np.array([w[0] for w in worst])
print(f"sibling spread (max over the stamps) over all {len(worst)} pixel ranges: p50 {np.median(sp_all):.1f} us, p90 {np.percentile(sp_all, 90):.1f}, max {sp_all.max():.1f}")
kl = t[:, 5] - t[:, 1]
print(f"k-loop of the first segment: p50 {np.median(kl):.1f} us, min {kl.min():.1f}, max {kl.max():.1f}")
