"""Is a step bound by the host's launch rate or by the device?  Issue K steps back to back and stamp the host clock when the LAST launch has been
enqueued (before any synchronisation) and again when the device has drained: host_ms = enqueue time per step, step_ms = wall per step.
host_ms ~ step_ms: the device waits for the host (launch-bound); host_ms << step_ms: the host runs ahead, the device is the limit.
    python tools/host_probe.py [--size 128] [--dtype bf16] [--workload dcpt|naf]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--workload", default="dcpt")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--sleep-before-opt-us", type=int, default=0, help="naf: host sleep between backward and optimizer.step() -- if the step time grows by it, the host is on the critical path there")
a = ap.parse_args()
import bench_extra as BX
from basicsr.models import build_model
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_
dev = torch.device("cuda:0")
naf = dict(BX.NAF, act_dtype=a.dtype)
B, S = 32, a.size
g = torch.Generator(device=dev).manual_seed(1)
if a.workload == "dcpt":
    opt = dict(name="b", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True, hook_names="decoder",
               network_g=dict(type="NAFNetBaseline", **naf),
               network_dc=dict(type="PromptIR_NoImg_DC", feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10, act_dtype="bf16" if a.dtype != "fp32" else "fp32"),
               path=dict(), train=dict(pixel_opt=dict(type="L1Loss"), classify_opt=dict(type="CrossEntropyLoss"),
                                       optim_g=dict(type="AdamW", lr=1e-4, fused=True), optim_dc=dict(type="AdamW", lr=1e-4, fused=True)))
    m = build_model(opt); fill_module_(m.net_g); fill_module_(m.net_dc)
    m.feed_data({"lq": torch.rand((B, 3, S, S), generator=g, device=dev), "gt": torch.rand((B, 3, S, S), generator=g, device=dev),
                 "dataset_idx": torch.randint(0, 10, (B,), generator=g, device=dev)})
    step = lambda: m.optimize_parameters(1)
else:
    from dcpt_amd.optim import FusedAdamW
    net = fill_module_(build_network(dict(type="NAFNetBaseline", **naf))).to(dev)
    optm = FusedAdamW(net.parameters(), lr=1e-4)
    lq = torch.rand((B, 3, S, S), generator=g, device=dev); gt = torch.rand_like(lq)
    def step():
        optm.zero_grad(set_to_none=True); (net(lq) - gt).abs().mean().backward()
        if a.sleep_before_opt_us: time.sleep(a.sleep_before_opt_us * 1e-6)
        optm.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{a.workload} {a.dtype} {S}x{S}: host enqueue {1e3 * (t1 - t0) / a.steps:.2f} ms/step, wall {1e3 * (t2 - t0) / a.steps:.2f} ms/step, device still busy after the last enqueue for {1e3 * (t2 - t1):.1f} ms")
