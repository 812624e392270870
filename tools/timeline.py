"""Phase timeline of one NT-GEMM launch (diagnostic build: tools/build_variant.sh tl -DDCPT_TIMELINE;
run with DCPT_TOOL_LIB=experiments/lib/libdcpt_hip_tl.so).  Stamps are the 100 MHz wall clock (10 ns)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF, _lib
dev = torch.device('cuda:0')
B, H, W = (int(v) for v in os.environ.get('TL_BHW', '32,32,32').split(','))
Ci, Co = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last); w = torch.randn(Co, Ci, 1, 1, device=dev)
lib = _lib.load()
rd = lib.dcpt_timeline_read; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int]
nblk = min(1 << 15, (B * H * W // 128) * ((Co + 127) // 128))
with torch.no_grad():
    for _ in range(200): DF.conv_nobias(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); DF.conv_nobias(x, w); e1.record(); torch.cuda.synchronize()
buf = np.zeros((nblk, 12), dtype=np.uint64)
assert rd(buf.ctypes.data, nblk) == 0
t = buf[:, :4].astype(np.int64); t0 = t[:, 0].min(); t = (t - t0) * 10e-3  # us
hw = buf[:, 4].astype(np.int64); xcc = buf[:, 5].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7   # gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
slot = xcc * 1000 + se * 100 + sh * 20 + cu
print(f"shape M={B*H*W} K={Ci} N={Co}  blocks={nblk}  event time {e0.elapsed_time(e1)*1e3:.1f} us   span of stamps {t[:,3].max():.1f} us   distinct CUs {len(set(slot))}")
def st(name, v): print(f"  {name:22s} min {v.min():7.2f}  p50 {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f} us")
order = np.argsort(t[:, 0]); first = order[:nblk // 2] if nblk > 512 else order; second = order[nblk // 2:] if nblk > 512 else order[:0]
for nm, idx in (("round 1", first), ("round 2", second)):
    if len(idx) == 0: continue
    print(nm, len(idx), "blocks")
    st("start", t[idx, 0]); st("prologue", t[idx, 1] - t[idx, 0]); st("main loop", t[idx, 2] - t[idx, 1]); st("epilogue", t[idx, 3] - t[idx, 2]); st("end", t[idx, 3])
# per-CU: idle gaps = time with <2 resident blocks
tot = t[:, 3].max(); occ = []
for s in set(slot):
    idx = np.where(slot == s)[0]
    occ.append((t[idx, 3] - t[idx, 0]).sum() / (2 * tot))
occ = np.array(occ); print(f"per-CU block-slot occupancy (2 slots): mean {occ.mean()*100:.1f}%  min {occ.min()*100:.1f}%  max {occ.max()*100:.1f}%   blocks/CU min {min(np.bincount(slot)[np.bincount(slot)>0])} max {np.bincount(slot).max()}")
ml = (t[:, 2] - t[:, 1]); print(f"main-loop share of block lifetime: {(ml.sum() / (t[:,3]-t[:,0]).sum())*100:.1f}%;  main-loop us/k-tile {np.median(ml) / (Ci/32):.3f} (MFMA-only for 2 co-resident waves/SIMD = {2*64*64/2.4e3:.3f} us at 2.4 GHz)")
cyc = (buf[:, 7].astype(np.int64) - buf[:, 6].astype(np.int64)); print(f"main loop shader cycles/k-tile: p50 {np.median(cyc)/(Ci/32):.0f} (MFMA-only 8192); implied clock {np.median(cyc)/np.median(ml)/1e3:.2f} GHz")
nk = Ci / 32
print(f"wave-0 per k-tile cycles: issue+MFMA+lstore {np.median(buf[:,8].astype(np.int64))/nk:.0f}  vmcnt(0) wait {np.median(buf[:,9].astype(np.int64))/nk:.0f}  barrier wait {np.median(buf[:,10].astype(np.int64))/nk:.0f}   (p90 vm {np.percentile(buf[:,9].astype(np.int64),90)/nk:.0f}, barrier {np.percentile(buf[:,10].astype(np.int64),90)/nk:.0f})")
if os.environ.get("TL_DUMP"):
    wid = hw & 0xf; simd = (hw >> 4) & 3
    for s in sorted(set(slot))[:6]:
        idx = np.where(slot == s)[0]
        print("CU", s, [(int(i), int(wid[i]), int(simd[i]), round(float(t[i, 0]), 1), round(float(t[i, 3]), 1)) for i in idx[np.argsort(t[idx, 0])]])
if os.environ.get("TL_XCD"):
    r1 = first
    for x in sorted(set(xcc)):
        idx = r1[xcc[r1] == x]
        print(f"  XCD {x}: n={len(idx)}  main loop p50 {np.median(ml[idx]):.1f}  min {ml[idx].min():.1f} max {ml[idx].max():.1f};  by SE: " +
              " ".join(f"{np.median(ml[idx[se[idx]==s_]]):.1f}" for s_ in sorted(set(se[idx]))))
    # fastest / slowest CUs
    cu_ml = {s: ml[np.where(slot == s)[0]].mean() for s in set(slot)}
    srt = sorted(cu_ml.items(), key=lambda kv: kv[1])
    print("  fastest CUs", [(k, round(v, 1)) for k, v in srt[:6]], " slowest", [(k, round(v, 1)) for k, v in srt[-6:]])
    # does wave slot matter?
    wid = hw & 0xf
    for wv in sorted(set(wid)):
        print(f"  wave_id {wv}: main loop p50 {np.median(ml[wid == wv]):.1f} n={np.sum(wid == wv)}")
