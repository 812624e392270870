"""GEMM classes (kernel class, M, N, K) of one step of a secondary workload, by HIP events around every launch (libdcpt_hip's prof.hip):
launches, total ms, us per launch, algorithmic TF/s and GB/s of each class.   python tools/gemm_classes.py restormer [--side-stream 0]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("workload", choices=["restormer"])
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=128)
a = ap.parse_args()
import __graft_entry__ as G
G.build()
from bench import prof_class_name
from dcpt_amd import _lib, functional as DF
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd.optim import FusedAdamW
from basicsr.archs import build_network
dev = torch.device("cuda", 0)
lib = _lib.load()
DF.set_restormer_save("balanced")
net = fill_module_(build_network(dict(type="Restormer"))).to(dev)
optm = FusedAdamW(net.parameters(), lr=1e-4)
lq = torch.rand((a.batch, 3, a.size, a.size), device=dev); gt = torch.rand_like(lq)
def step():
    optm.zero_grad(set_to_none=True)
    (net(lq) - gt).abs().mean().backward()
    optm.step()
for _ in range(2): step()
torch.cuda.synchronize()
lib.dcpt_prof_enable(1)
step(); torch.cuda.synchronize()
buf = (ctypes.c_double * (8 * 512))()
n = lib.dcpt_prof_read(buf, 512)
lib.dcpt_prof_enable(0)
rows = [(buf[i * 8 + 5], prof_class_name(int(buf[i * 8])), int(buf[i * 8 + 1]), int(buf[i * 8 + 2]), int(buf[i * 8 + 3]), int(buf[i * 8 + 4]), buf[i * 8 + 6], buf[i * 8 + 7]) for i in range(n)]
tot = sum(r[0] for r in rows)
print(f"{n} classes, {tot:.1f} ms of bracketed launches in one step")
for ms, name, M, N, K, cnt, fl, by in sorted(rows, reverse=True)[:60]:
    us = 1e3 * ms / max(cnt, 1)
    print(f"{ms:8.2f} ms  x{cnt:4d}  {us:8.1f} us  {name:36s} M={M:8d} N={N:5d} K={K:5d}  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e6:7.0f} GB/s  "
          f"bound mfma {fl/cnt/125e12*1e6:6.1f} us / hbm {by/cnt/5e12*1e6:6.1f} us")
