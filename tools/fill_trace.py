"""Where do the small torch kernels inside the timed step come from?  One step of the default bench workload (or --dcpt) under
torch.profiler with Python stacks; prints, per torch kernel name, the launch count and the most frequent Python call sites."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--match", default="Fill,copyBuffer,elementwise,Memcpy,Memset,at::native")
ap.add_argument("--dcpt", action="store_true", help="the all-bf16 DCPT step (configs[2]) instead of the NAFNet training step")
ap.add_argument("--size", type=int, default=128)
a = ap.parse_args()
import __graft_entry__ as G
G.build()
from dcpt_amd.keyed_init import fill_module_
dev = torch.device("cuda", 0)
naf = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
if a.dcpt:
    from basicsr.models import build_model
    opt = dict(name="b", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               hook_names="decoder", network_g=dict(type="NAFNetBaseline", act_dtype="bf16", **naf),
               network_dc=dict(type="PromptIR_NoImg_DC", feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10, act_dtype="bf16"),
               path=dict(), train=dict(pixel_opt=dict(type="L1Loss"), classify_opt=dict(type="CrossEntropyLoss"),
                                       optim_g=dict(type="AdamW", lr=1e-4, fused=True), optim_dc=dict(type="AdamW", lr=1e-4, fused=True)))
    m = build_model(opt)
    fill_module_(m.net_g); fill_module_(m.net_dc)
    S = a.size
    m.feed_data({"lq": torch.rand((a.batch, 3, S, S), device=dev), "gt": torch.rand((a.batch, 3, S, S), device=dev),
                 "dataset_idx": torch.randint(0, 10, (a.batch,), device=dev)})
    def step(): m.optimize_parameters(1)
else:
    from basicsr.archs import build_network
    net = build_network(dict(type="NAFNetBaseline", **naf))
    fill_module_(net, seed=0)
    net = net.to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.9), weight_decay=0.0, fused=True)
    lq = torch.rand((a.batch, 3, 256, 256), device=dev); gt = torch.rand_like(lq)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = (net(lq) - gt).abs().mean(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
keys = a.match.split(",")
ops = collections.Counter(); sites = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and any(k.lower() in ev.name.lower() for k in ("fill", "zero", "copy_", "add", "mul", "div", "ones", "full")):
        st = [s for s in (ev.stack or []) if "/torch/" not in s and "<built-in" not in s][:3]
        ops[ev.name] += 1; sites[ev.name][" <- ".join(st) or "(no python frame: autograd engine)"] += 1
# device kernels matching --match, attributed to the CPU-side event that launched them (FunctionEvent.kernels)
kk = collections.Counter(); ksites = collections.defaultdict(collections.Counter)
for ev in prof.events():
    for k in (getattr(ev, "kernels", None) or []):
        if any(m.lower() in k.name.lower() for m in keys):
            short = k.name[:110]
            kk[short] += 1
            st = [s for s in (ev.stack or []) if "/torch/" not in s and "<built-in" not in s][:3]
            ksites[short][ev.name + " " + str(getattr(ev, "input_shapes", ""))[:80] + " @ " + (" <- ".join(st) or "-")] += 1
for name, n in kk.most_common(8):
    print(f"{n:6d}  KERNEL {name}")
    for s, c in ksites[name].most_common(24): print(f"        {c:5d}  {s}")
for name, n in ops.most_common(20):
    print(f"{n:6d}  {name}")
    for s, c in sites[name].most_common(4): print(f"        {c:5d}  {s}")
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=70))
