"""GPU: the C-ABI gradient all-reduce ``dcpt_allreduce_flat`` (include/dcpt_hip.h; reference base_model.py:108-115 / :448)
on a caller-provided RCCL communicator: a 1-rank communicator on the single test GPU, and two ranks when the node has two."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_allreduce_flat_one_rank():
    from dcpt_amd import _lib, comm

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    c = comm.RcclComm(1, 0, comm.unique_id())
    try:
        for n in (1, 3, 4, 1023, 1 << 20):
            buf = torch.randn(n, device=dev)
            want = buf.clone()
            comm.allreduce_flat_(buf, c, mean=True)     # world 1: sum == identity, scale 1
            torch.cuda.synchronize()
            assert torch.equal(buf, want), n
        # explicit scale through the raw entry point
        lib = _lib.load()
        y = torch.arange(10, device=dev, dtype=torch.float32)
        _lib.check(lib.dcpt_allreduce_flat(y.data_ptr(), y.numel(), c.handle, 0.5, torch.cuda.current_stream().cuda_stream), "allreduce")
        torch.cuda.synchronize()
        assert torch.equal(y.cpu(), torch.arange(10, dtype=torch.float32) * 0.5)
        assert lib.dcpt_allreduce_flat(None, 4, c.handle, 1.0, None) != 0 and b"null" in lib.dcpt_last_error()
        with pytest.raises(_lib.DcptHipError):
            comm.allreduce_flat_(torch.zeros(4), c)
    finally:
        c.close()


def _worker(rank, world, uid_path, out_path):
    import time

    from dcpt_amd import comm

    torch.cuda.set_device(rank)
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(comm.unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.05)
    c = comm.RcclComm(world, rank, open(uid_path, "rb").read())
    g = torch.Generator().manual_seed(100 + rank)
    buf = torch.randn(100003, generator=g).cuda()
    comm.allreduce_flat_(buf, c, mean=True)
    torch.cuda.synchronize()
    torch.save(buf.cpu(), f"{out_path}.{rank}")
    c.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the node")
def test_allreduce_flat_two_ranks(tmp_path):
    import torch.multiprocessing as mp

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, args=(2, str(tmp_path / "uid"), str(tmp_path / "out")), nprocs=2, join=True)
    want = sum(torch.randn(100003, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)) / 2
    for r in range(2):
        got = torch.load(f"{tmp_path / 'out'}.{r}")
        assert torch.allclose(got, want, rtol=0, atol=1e-6), r


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="checks the refusal on a single-GPU box")
def test_bench_refuses_more_gpus_than_present():
    """`python bench.py --gpus 2` without a launcher spawns its own ranks; with fewer devices than asked it must fail loudly
    instead of printing an n_gpus=1 line (round-1 verdict, weak #5)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "only 1 device" in (p.stdout + p.stderr) and '"n_gpus"' not in p.stdout


# ------------------------------------------------------------------------------------------------------------------------
TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])


def _ddp_two_rank_worker(rank, world, port, out):
    """one rank of a 2-rank data-parallel run of the HIP network: both ranks share cuda:0 (there is one GPU), the collective goes over gloo"""
    import sys

    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from basicsr.archs import build_network
    from basicsr.archs.nafnet_arch import NAFBlock
    from dcpt_amd import ddp as dcpt_ddp, functional as DF
    from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
    from oracle import nafnet_oracle as O

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net = build_network(dict(type="NAFNetBaseline", **TINY))
    net.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
    net = net.to(dev)
    model = dcpt_ddp.prepare(DistributedDataParallel(net, device_ids=[0], bucket_cap_mb=0.02, gradient_as_bucket_view=True))
    x = keyed_input(f"ddp2r.x{rank}", (4, 3, 32, 32)).to(dev)
    gw = keyed_input("ddp2r.gw", (4, 3, 32, 32), lo=-1, hi=1).to(dev)
    opt = torch.optim.AdamW(net.parameters(), lr=0.0, fused=True)
    nblock = sum(len(list(m.parameters())) for m in net.modules() if isinstance(m, NAFBlock))
    hits = []
    for it in range(4):
        h0 = DF._grad_buffers.hits
        opt.zero_grad(set_to_none=True)
        (model(x) * gw).sum().backward()
        torch.cuda.synchronize()
        hits.append(DF._grad_buffers.hits - h0)
        grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
        opt.step()
    torch.save({"grads": grads, "hits": hits, "nblock": nblock}, os.path.join(out, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_two_ranks_one_gpu_gloo(tmp_path):
    """SURVEY 8e on the hardware there is: TWO data-parallel ranks of the HIP network (DDP as BaseModel.model_to_device and bench.py set it
    up: bucket views, built-in all-reduce hook, the blocks' gradients written straight into the bucket views -- dcpt_amd/ddp.py), both on
    cuda:0, the all-reduce over gloo.  Every rank ends with the MEAN of the two ranks' single-process gradients, in the steady state every
    NAFBlock gradient is produced in place (no copies into the buckets), and the two ranks agree bit for bit."""
    import torch.multiprocessing as mp

    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
    from oracle import nafnet_oracle as O

    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_ddp_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"g{i}.pt")) for i in range(2)]
    assert r[0]["hits"][2:] == [r[0]["nblock"]] * 2 == r[1]["hits"][2:], (r[0]["hits"], r[0]["nblock"])
    dev = torch.device("cuda:0")
    gw = keyed_input("ddp2r.gw", (4, 3, 32, 32), lo=-1, hi=1).to(dev)
    local = []
    for rank in range(2):
        net = build_network(dict(type="NAFNetBaseline", **TINY))
        net.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0), strict=True)
        net = net.to(dev)
        (net(keyed_input(f"ddp2r.x{rank}", (4, 3, 32, 32)).to(dev)) * gw).sum().backward()
        local.append({k: p.grad.detach().cpu() for k, p in net.named_parameters()})
    for k in local[0]:
        want = (local[0][k] + local[1][k]) / 2
        assert torch.equal(r[0]["grads"][k], r[1]["grads"][k]), k
        assert torch.allclose(r[0]["grads"][k], want, rtol=1e-6, atol=1e-7 * float(want.abs().max()) + 1e-12), k
