"""Does replaying the whole training step as one HIP graph beat stream launches?  (single stream: the library keeps
graph captures on the capturing stream)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401  (DCPT_TOOL_LIB)
from basicsr.archs import build_network
from dcpt_amd import _lib
from dcpt_amd.keyed_init import fill_module_
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
dev = torch.device('cuda:0')
lib = _lib.load()
ACT = sys.argv[1] if len(sys.argv) > 1 else "fp32"   # python tools/graph_probe.py [bf16]
net = build_network(dict(type="NAFNetBaseline", act_dtype=ACT, **CFG)); fill_module_(net, seed=0); net = net.to(dev)
opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0, fused=True, capturable=True)
lq = torch.rand((32, 3, 256, 256), device=dev); gt = torch.rand((32, 3, 256, 256), device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    loss = (net(lq) - gt).abs().mean()
    loss.backward()
    opt.step()
    return loss
def timed(fn, n=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
lib.dcpt_set_side_stream(0)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
print("eager, single stream: %.2f ms" % timed(step))
lib.dcpt_set_side_stream(1)
print("eager, side stream:   %.2f ms" % timed(step))
lib.dcpt_set_side_stream(0)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    loss = (net(lq) - gt).abs().mean()
    loss.backward()
    opt.step()
print("graph replay:         %.2f ms" % timed(g.replay), " loss", float(loss))
