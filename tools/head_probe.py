"""The classifier head alone (PromptIR_NoImg_DC([64,128,256,512]), bf16 activations): forward + backward on feature maps of the DCPT step's
sizes (B = 32, 256 x 256 at stage 0), for a per-kernel table of the head by itself:
    tools/kernel_table.sh out.txt 5 python tools/head_probe.py [--size 256] [--batch 32]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
if os.environ.get("DCPT_TOOL_LIB"):   # a variant build (tools/build_variant.sh) instead of the product library
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _variant  # noqa: F401
import __graft_entry__ as G
G.build()
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_

dev = torch.device("cuda", 0)
dims = [64, 128, 256, 512]
net = build_network(dict(type="PromptIR_NoImg_DC", feature_dims=dims, num_res_blocks=2, num_classes=10, act_dtype="bf16")).to(dev)
fill_module_(net)
g = torch.Generator(device=dev).manual_seed(0)
feats = [(torch.rand((a.batch, c, a.size >> i, a.size >> i), generator=g, device=dev) - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
         for i, c in enumerate(dims)]
labels = torch.randint(0, 10, (a.batch,), generator=g, device=dev)


def step():
    net.zero_grad(set_to_none=True)
    fd = [f.detach().requires_grad_(True) for f in feats]
    loss = torch.nn.functional.cross_entropy(net(None, fd), labels)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps - 2):
    step()
torch.cuda.synchronize()
print(f"head alone, B={a.batch}, {a.size}x{a.size}: {(time.perf_counter() - t0) / (a.steps - 2) * 1e3:.2f} ms per forward + backward")
