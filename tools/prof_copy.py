import os, sys, torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
dev = torch.device('cuda:0')
net = fill_module_(build_network(dict(type="NAFNetBaseline", **CFG))).to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True)
lq = torch.rand((32, 3, 256, 256), device=dev); gt = torch.rand((32, 3, 256, 256), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    (net(lq) - gt).abs().mean().backward()
    opt.step()
for _ in range(2): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if any(k in e.key for k in ("copy", "Memcpy", "contiguous", "clone", "fill", "zero", "add"))]
for e in sorted(rows, key=lambda e: -e.count)[:14]:
    print(f"{e.key[:60]:60s} count {e.count:5d}  cuda {e.device_time_total/1e3:8.3f} ms cpu {e.cpu_time_total/1e3:8.3f} ms")
