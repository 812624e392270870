"""Phase timeline of one TN (weight-gradient) GEMM launch (diagnostic build: tools/build_variant.sh tl "gemm_nt.hip gemm_tn.hip"
-DDCPT_TIMELINE=1; run with DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_tl.so)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF, _lib
dev = torch.device('cuda:0')
B, H, W = 32, 32, 32
Ci, Co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
lib = _lib.load(); lib.dcpt_set_side_stream(0)
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, 1, 1, device=dev, requires_grad=True)
go = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
rd = lib.dcpt_timeline_read_tn; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int]
for _ in range(30):
    w.grad = None; DF.conv_nobias(x, w).backward(go)
torch.cuda.synchronize()
nblk = 512
buf = np.zeros((nblk, 8), dtype=np.uint64); assert rd(buf.ctypes.data, nblk) == 0
t = buf[:, :4].astype(np.int64); t = (t - t[:, 0].min()) * 10e-3
cyc = buf[:, 4:8].astype(np.int64)
M = B * H * W; tiles = ((Co + 127) // 128) * ((Ci + 127) // 128); splits = 512 // tiles; steps = M // splits // 32
def st(n, v): print(f"  {n:12s} min {v.min():7.2f} p50 {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f} us")
print(f"TN M={M} N={Co} K={Ci}: {tiles} tiles x {splits} splits, {steps} pixel tiles of 32 rows per block; span {t[:,3].max():.1f} us")
st("start", t[:, 0]); st("prologue", t[:, 1] - t[:, 0]); st("main loop", t[:, 2] - t[:, 1]); st("epilogue", t[:, 3] - t[:, 2]); st("end", t[:, 3])
ml = cyc[:, 2] - cyc[:, 1]
print(f"main loop shader cycles per pixel tile: p50 {np.median(ml)/steps:.0f} (MFMA-only for the 2 co-resident waves/SIMD: 8192)")
