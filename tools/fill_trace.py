"""Where do the small torch kernels inside the timed step come from?  One step of the default bench workload (or --dcpt) under
torch.profiler with Python stacks; prints, per torch kernel name, the launch count and the most frequent Python call sites."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--match", default="Fill,copyBuffer,elementwise,Memcpy,Memset")
a = ap.parse_args()
import __graft_entry__ as G
G.build()
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_
dev = torch.device("cuda", 0)
net = build_network(dict(type="NAFNetBaseline", img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1]))
fill_module_(net, seed=0)
net = net.to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0, fused=True)
lq = torch.rand((a.batch, 3, 256, 256), device=dev); gt = torch.rand_like(lq)
def step():
    opt.zero_grad(set_to_none=True)
    loss = (net(lq) - gt).abs().mean(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
keys = a.match.split(",")
ops = collections.Counter(); sites = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and any(k.lower() in ev.name.lower() for k in ("fill", "zero", "copy_", "add", "mul", "div", "ones", "full")):
        st = [s for s in (ev.stack or []) if "/torch/" not in s and "<built-in" not in s][:3]
        ops[ev.name] += 1; sites[ev.name][" <- ".join(st) or "(no python frame: autograd engine)"] += 1
for name, n in ops.most_common(20):
    print(f"{n:6d}  {name}")
    for s, c in sites[name].most_common(4): print(f"        {c:5d}  {s}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
