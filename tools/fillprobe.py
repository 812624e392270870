import sys, torch
sys.path.insert(0, '.')
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
dev = torch.device('cuda:0')
net = fill_module_(build_network(dict(type="NAFNetBaseline", **CFG))).to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True)
lq = torch.rand(4, 3, 256, 256, device=dev); gt = torch.rand(4, 3, 256, 256, device=dev)
def step():
    opt.zero_grad(set_to_none=True); out = net(lq); loss = (out - gt).abs().mean(); loss.backward(); opt.step()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.count >= 30]
for r in sorted(rows, key=lambda r: -r[1])[:25]: print(r)
