"""What the DDP wrap costs on ONE rank (RCCL group of one): the headline step unwrapped / DDP as torch ships it / + per-bucket divide hook /
+ the blocks' gradients written into the bucket views (dcpt_amd/ddp.py); and the copy-type kernels per step of each.
    python tools/ddp_probe.py"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP
from basicsr.archs import build_network
from dcpt_amd import ddp as dcpt_ddp, functional as DF
from dcpt_amd.keyed_init import fill_module_

dev = torch.device("cuda:0")
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
g = torch.Generator(device=dev).manual_seed(1)
lq = torch.rand((32, 3, 256, 256), generator=g, device=dev); gt = torch.rand((32, 3, 256, 256), generator=g, device=dev)


def run(mode):
    net = fill_module_(build_network(dict(type="NAFNetBaseline", **CFG)), seed=0).to(dev)
    model = net
    if mode != "bare":
        model = DDP(net, device_ids=[0], bucket_cap_mb=64, gradient_as_bucket_view=True)
        if mode == "hook":
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            model.register_comm_hook(None, default_hooks.allreduce_hook)
        if mode == "builtin":
            model._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
        if mode == "views":
            dcpt_ddp.prepare(model)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        (model(lq) - gt).abs().mean().backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    cnt = {}
    for e in prof.key_averages():
        if e.key in ("aten::copy_", "aten::clone", "aten::mul", "aten::mul_", "aten::div_", "aten::div", "aten::empty_like", "aten::add_", "aten::_foreach_div_"):
            cnt[e.key] = e.count
    print(f"{mode:6s} {ms:8.2f} ms/step   {cnt}", flush=True)
    del net, model, opt
    torch.cuda.empty_cache()


for m in ("bare", "ddp", "hook", "builtin", "views", "bare"):   # views = dcpt_ddp.prepare: built-in hook + gradients written into the bucket views
    run(m)
dist.destroy_process_group()
