"""A/B inside one process: bf16 NAFNet-64 step (B = 32, 256^2) with and without the per-block packed-weights cache."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr.archs import build_network
from dcpt_amd import functional as DF
from dcpt_amd.keyed_init import fill_module_
NAF = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1], act_dtype="bf16")
dev = torch.device("cuda:0")
net = fill_module_(build_network(dict(type="NAFNetBaseline", **NAF))).to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0, fused=True)
g = torch.Generator(device=dev).manual_seed(1)
lq = torch.rand((32, 3, 256, 256), generator=g, device=dev); gt = torch.rand((32, 3, 256, 256), generator=g, device=dev)
orig = DF.nafblock_bf16
def step():
    opt.zero_grad(set_to_none=True); (net(lq) - gt).abs().mean().backward(); opt.step()
def timed(n=8):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    DF.nafblock_bf16 = orig
    a = timed()
    DF.nafblock_bf16 = lambda inp, params, packed=None: orig(inp, params, None)
    b = timed()
    print(f"round {rnd}: packed cache {a:.2f} ms/step, per-call packs {b:.2f} ms/step")
