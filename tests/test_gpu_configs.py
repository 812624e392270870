"""BASELINE.json configurations at their REAL sizes on the HIP path (round-1 verdict: `configs[2]` and `configs[4]` were only
exercised with tiny nets), the data-parallel wrap on the device, and the PSNR acceptance gate of `north_star`.

* configs[4] -- 2048 x 2048 image, `tile: {infer_size: 512, tile_pad: 16}`, NAFNet-64: pasted tiles vs the ORACLE run on the same
  padded crops (reference basicsr/models/sr_model.py:273-361), and batched-by-shape == one-tile-at-a-time.
* configs[2] (fp32 arithmetic) -- `DCPTModel.optimize_parameters` with NAFNet-64 + `PromptIR_NoImg_DC(feature_dims=[64,128,256,512])`,
  B = 32, 128 x 128: directional derivative of the whole step (reference ...pretrain_model.py:133-169).
* M3 -- the tiny HIP NAFNet wrapped by `BaseModel.model_to_device` in DistributedDataParallel on a 1-rank `nccl` (= RCCL) group:
  gradients bit-identical to the unwrapped run (reference base_model.py:108-115).
* PSNR gate -- NAFNet-64 TRAINED in the module fixture (240 iterations on synthetic sigma = 25 pairs, no layer scaled), 4 held-out
  256 x 256 pairs: |PSNR(HIP) - PSNR(oracle)| <= 0.01 dB, |SSIM diff| <= 1e-4 with the uint8 rounding of reference
  basicsr/metrics/psnr_ssim.py:47-75.
"""
import os
import socket

import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
from oracle import dc_oracle as D
from oracle import nafnet_oracle as O

pytestmark = pytest.mark.gpu

TINY = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
DC_FULL = dict(feature_dims=[64, 128, 256, 512], num_res_blocks=2, num_classes=10)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dcpt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _smooth_pair(index, size=256, sigma=25.0 / 255.0):
    """a seeded (gt, lq) pair that looks like an image: smooth random field + Gaussian noise"""
    g = torch.Generator().manual_seed(1000 + index)
    low = torch.rand((1, 3, size // 16, size // 16), generator=g)
    gt = torch.nn.functional.interpolate(low, size=(size, size), mode="bicubic", align_corners=False)[0].clamp(0, 1)
    lq = (gt + sigma * torch.randn((3, size, size), generator=g)).clamp(0, 1)
    return gt, lq


# ------------------------------------------------------------------------------------------------------------------------
def test_tiled_2k_inference_vs_oracle(dev):
    """BASELINE.json configs[4]."""
    from basicsr.models import build_model

    opt = dict(name="t", model_type="SRModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=False,
               network_g=dict(type="NAFNetBaseline", window_size=16, **FULL), path=dict(), tile=dict(infer_size=512, tile_pad=16),
               val=dict(save_img=False))
    m = build_model(opt)
    sd = keyed_state_dict(O.nafnet_param_shapes(**FULL), seed=0)
    m.net_g.load_state_dict(sd, strict=True)
    img = torch.rand((1, 3, 2048, 2048), generator=torch.Generator().manual_seed(2048))
    m.feed_data({"lq": img})
    m.pre_test()
    assert m.lq.shape == (1, 3, 2048, 2048)
    m.test_tile()
    m.post_test()
    got = m.output.cpu()
    assert got.shape == (1, 3, 2048, 2048) and bool(torch.isfinite(got).all())

    # (1) the reference's loop (sr_model.py:291-361), one tile at a time on the device: same tiles, same paste arithmetic
    size, pad = 512, 16
    seq = torch.zeros_like(img)
    lq = img.to(dev)
    with torch.no_grad():
        for ty in range(4):
            for tx in range(4):
                x0, y0 = tx * size, ty * size
                x1, y1 = min(x0 + size, 2048), min(y0 + size, 2048)
                xp0, yp0, xp1, yp1 = max(x0 - pad, 0), max(y0 - pad, 0), min(x1 + pad, 2048), min(y1 + pad, 2048)
                out = m.net_g(lq[:, :, yp0:yp1, xp0:xp1].contiguous())
                seq[:, :, y0:y1, x0:x1] = out[:, :, y0 - yp0:y0 - yp0 + size, x0 - xp0:x0 - xp0 + size].cpu()
    # batched-by-shape vs sequential: every output pixel is the same chain of fp32 operations; only the per-image pooling
    # partial sums are grouped differently for a different batch size (the row-range split of the depthwise kernel follows the
    # batch to fill the chip), which perturbs SCA by an ulp and propagates through 36 blocks: 2e-5 instead of bit equality
    err = _relerr(got, seq)
    assert err <= 2e-5, f"batched tiles vs sequential tiles: {err:.3e} (bit-equal: {torch.equal(got, seq)})"

    # (2) the ORACLE on the padded crop of one corner tile (528 x 528) and one interior tile (544 x 544)
    P = {k: v for k, v in sd.items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    for ty, tx in ((0, 0), (1, 2)):
        x0, y0 = tx * size, ty * size
        xp0, yp0, xp1, yp1 = max(x0 - pad, 0), max(y0 - pad, 0), min(x0 + size + pad, 2048), min(y0 + size + pad, 2048)
        with torch.no_grad():
            ref, _ = O.nafnet_forward(img[:, :, yp0:yp1, xp0:xp1].contiguous(), P)
        want = ref[:, :, y0 - yp0:y0 - yp0 + size, x0 - xp0:x0 - xp0 + size]
        e = _relerr(got[:, :, y0:y0 + size, x0:x0 + size], want)
        assert e <= 1e-3, f"tile ({ty},{tx}) vs oracle: {e:.3e}"


@pytest.mark.parametrize("act_dtype", ["fp32", "bf16"])
def test_tile_batches_on_several_streams_equal_one_stream(dev, act_dtype):
    """configs[4]: ``test_tile`` runs its shape classes on ``tile.streams`` HIP streams (default 2).  Same batches, same kernels, disjoint
    output regions: the result is bit-identical to the single-stream run, repeatedly (a race on the shared weight packs, a workspace or
    the output would show as run-to-run differences), and the packed weights made on one stream are waited for by the others."""
    from basicsr.models import build_model
    from dcpt_amd import functional as DF

    outs = {}
    for streams in (1, 4):
        opt = dict(name="t", model_type="SRModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=False,
                   network_g=dict(type="NAFNetBaseline", window_size=16, act_dtype=act_dtype, **FULL), path=dict(),
                   tile=dict(infer_size=512, tile_pad=16, streams=streams), val=dict(save_img=False))
        m = build_model(opt)
        m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**FULL), seed=0), strict=True)
        img = torch.rand((1, 3, 1536, 1280), generator=torch.Generator().manual_seed(7))
        runs = []
        for rep in range(3):
            if rep == 1:
                DF.invalidate_packed_weights()   # the next first batch re-packs on ITS stream while the others start behind it
            m.feed_data({"lq": img})
            m.pre_test()
            m.test_tile()
            m.post_test()
            runs.append(m.output.cpu())
        # (the split-operand GEMM mode shares one scratch buffer process-wide: one stream there)
        assert len(m._tile_streams(9, m.net_g)) == (4 if streams == 4 and DF.get_gemm_precision() == "fp32" else 0)
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
        outs[streams] = runs[0]
        if streams == 4:
            # a network built for the split-operand mode keeps ONE stream also behind a DataParallel / DDP wrapper (which hides the
            # attribute the guard reads: round-4 advisor finding) -- the mode's scratch buffer is process-wide
            wrapped = torch.nn.DataParallel(m.net_g, device_ids=[m.lq.device.index or 0])
            assert len(m._tile_streams(9, wrapped)) == len(m._tile_streams(9, m.net_g))
            m.net_g.gemm_precision = "bf16x3"
            try:
                assert m._tile_streams(9, m.net_g) == [] and m._tile_streams(9, wrapped) == []
            finally:
                m.net_g.gemm_precision = None
    assert bool(torch.isfinite(outs[1]).all()) and torch.equal(outs[1], outs[4])


# ------------------------------------------------------------------------------------------------------------------------
def test_dcpt_step_full_size_directional_derivative(dev):
    """BASELINE.json configs[2] in fp32: the analytic gradients left in `.grad` by DCPTModel.optimize_parameters (lr = 0), contracted
    with a random direction in the parameter space of BOTH networks, equal the central difference of the step's loss
    (l_pix + l_classify) evaluated by forward passes only."""
    from basicsr.models import build_model

    B, S = 32, 128
    opt = dict(name="t", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               hook_names="decoder", network_g=dict(type="NAFNetBaseline", **FULL),
               network_dc=dict(type="PromptIR_NoImg_DC", **DC_FULL), path=dict(),
               train=dict(pixel_opt=dict(type="MSELoss", loss_weight=1.0, reduction="mean"),
                          classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0),
                          optim_g=dict(type="SGD", lr=0.0), optim_dc=dict(type="SGD", lr=0.0)))
    m = build_model(opt)
    m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**FULL), seed=0), strict=True)
    m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**DC_FULL), seed=0), strict=True)
    assert len(m.hooks) == 4
    gen = torch.Generator().manual_seed(21)
    gt = torch.rand((B, 3, S, S), generator=gen)
    lq = (gt + 0.1 * torch.randn((B, 3, S, S), generator=gen)).clamp(0, 1)
    labels = torch.randint(0, 10, (B,), generator=gen)
    m.feed_data({"lq": lq, "gt": gt, "dataset_idx": labels})
    m.optimize_parameters(1)
    log = m.get_current_log()

    def losses():   # the step's forward half (…pretrain_model.py:140-160), losses reduced in fp64
        with torch.no_grad():
            pix = m.net_g(m.gt, hook=False)
            m.hook_outputs = []
            m.net_g(m.lq, hook=True)
            cls = m.net_dc(m.lq, m.hook_outputs[::-1])
            m.hook_outputs = []
            l_pix = (pix.double() - m.gt.double()).pow(2).mean()
            l_cls = torch.nn.functional.cross_entropy(cls.double(), m.dataset_idx)
        return float(l_pix), float(l_cls)

    lp, lc = losses()
    assert abs(lp - log["l_pix"]) <= 1e-5 * max(1.0, abs(lp)) and abs(lc - log["l_classify"]) <= 1e-4 * max(1.0, abs(lc)), (lp, lc, log)
    params = [p for p in m.net_g.parameters()] + [p for p in m.net_dc.parameters()]
    assert all(p.grad is not None for p in params)
    dirs = [torch.randn(p.shape, generator=gen).to(dev) * p.detach().abs().mean().clamp_min(1e-3) for p in params]
    analytic = float(sum((p.grad.double() * d.double()).sum() for p, d in zip(params, dirs)))
    eps = 1e-3
    vals = []
    with torch.no_grad():
        for sign in (+1.0, -1.0):
            for p, d in zip(params, dirs):
                p.add_(d, alpha=sign * eps)
            vals.append(sum(losses()))
            for p, d in zip(params, dirs):
                p.sub_(d, alpha=sign * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    assert abs(analytic) > 1e-3, analytic
    assert abs(numeric - analytic) <= 3e-2 * abs(analytic), (numeric, analytic, lp, lc)


# ------------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_ddp_wrapped_hip_net_one_rank_rccl(dev):
    """reference base_model.py:108-115: the DDP wrap of the HIP network on an `nccl` (RCCL) process group of one rank -- bucket
    views, autograd hooks firing block by block and the all-reduce leave exactly the gradients of the unwrapped run."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel

    from basicsr.archs import build_network
    from basicsr.models.base_model import BaseModel

    sd = keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0)
    x = keyed_input("ddp.x", (4, 3, 32, 32)).to(dev)
    gw = keyed_input("ddp.gw", (4, 3, 32, 32), lo=-1, hi=1).to(dev)

    def run(wrap):
        net = build_network(dict(type="NAFNetBaseline", **TINY))
        net.load_state_dict(sd, strict=True)
        if wrap:
            bm = BaseModel(dict(num_gpu=1, is_train=True, dist=True, rank=0, world_size=1))
            model = bm.model_to_device(net)
            assert isinstance(model, DistributedDataParallel)
        else:
            model = net.to(dev)
        for _ in range(2):   # second iteration: DDP has rebuilt its buckets in gradient-arrival order
            model.zero_grad(set_to_none=True)
            y = model(x)
            (y * gw).sum().backward()
        torch.cuda.synchronize()
        bare = model.module if wrap else model
        return y.detach().clone(), {k: p.grad.detach().clone() for k, p in bare.named_parameters()}

    y0, g0 = run(False)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        t = torch.ones(8, device=dev)
        dist.all_reduce(t)
        assert float(t.sum()) == 8.0
        y1, g1 = run(True)
    finally:
        dist.destroy_process_group()
    assert torch.equal(y0, y1)
    assert set(g0) == set(g1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_ddp_bucket_order_and_zero_copy_gradients(dev):
    """Multi-GPU readiness that one GPU can show (reference base_model.py:108-115).  On a 1-rank RCCL group, the HIP network
    wrapped as BaseModel.model_to_device wraps it: (i) from DDP's re-bucketing on, the buckets are all-reduced in exactly the order in
    which the backward pass finishes the gradients -- ``ending`` first, ``intro`` last -- so the all-reduce of a bucket overlaps the
    backward kernels of the blocks before it by construction, not by luck; (ii) the fused blocks write their parameter gradients
    straight into the bucket views (dcpt_amd/ddp.py): after the optimizer steps that teach the views, every NAFBlock gradient of a
    step is produced in place and the wrap adds no per-parameter copy kernels; gradients stay bit-identical to the unwrapped run."""
    import torch.distributed as dist
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    from torch.nn.parallel import DistributedDataParallel

    from basicsr.archs import build_network
    from basicsr.archs.nafnet_arch import NAFBlock
    from dcpt_amd import ddp as dcpt_ddp, functional as DF

    sd = keyed_state_dict(O.nafnet_param_shapes(**TINY), seed=0)
    x = keyed_input("ddp2.x", (4, 3, 32, 32)).to(dev)
    gw = keyed_input("ddp2.gw", (4, 3, 32, 32), lo=-1, hi=1).to(dev)

    def grads_of(net):
        return {k: p.grad.detach().clone() for k, p in net.named_parameters()}

    net0 = build_network(dict(type="NAFNetBaseline", **TINY))
    net0.load_state_dict(sd, strict=True)
    net0 = net0.to(dev)
    (net0(x) * gw).sum().backward()
    g0 = grads_of(net0)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        net = build_network(dict(type="NAFNetBaseline", **TINY))
        net.load_state_dict(sd, strict=True)
        net = net.to(dev)
        names = {id(p): k for k, p in net.named_parameters()}
        model = DistributedDataParallel(net, device_ids=[torch.cuda.current_device()], bucket_cap_mb=0.02, gradient_as_bucket_view=True)
        buckets, ready = [], []

        def hook(state, bucket):
            buckets.append([names[id(p)] for p in bucket.parameters()])
            return default_hooks.allreduce_hook(state, bucket)

        dcpt_ddp.prepare(model, hook)
        for p in net.parameters():
            p.register_post_accumulate_grad_hook(lambda q: ready.append(names[id(q)]))
        opt = torch.optim.AdamW(net.parameters(), lr=0.0, fused=True)   # (lr = 0: the weights stay, its post-step hook teaches the views)
        nblock = sum(len(list(m.parameters())) for m in net.modules() if isinstance(m, NAFBlock))
        for it in range(4):
            buckets.clear()
            ready.clear()
            hits = DF._grad_buffers.hits
            opt.zero_grad(set_to_none=True)
            (model(x) * gw).sum().backward()
            torch.cuda.synchronize()
            if it >= 2:   # DDP re-bucketed after its first iteration; the views of the new buckets were learned after the second
                flat = [k for b in buckets for k in b]
                assert len(buckets) >= 4 and flat == ready, (len(buckets), flat[:6], ready[:6])
                assert ready[0].startswith("ending.") and ready[-1].startswith("intro."), (ready[0], ready[-1])
                assert DF._grad_buffers.hits - hits == nblock, (DF._grad_buffers.hits - hits, nblock)
                for m in net.modules():
                    if isinstance(m, NAFBlock):
                        for p in m.parameters():
                            assert p.grad.data_ptr() == p._dcpt_grad_view.data_ptr()
            g1 = grads_of(net)
            for k in g0:
                assert torch.equal(g0[k], g1[k]), (it, k)
            opt.step()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def trained_denoiser(dev):
    """A NAFNet-64 that really restores: trained HERE, on the HIP path, from the reference's default initialisation (beta = gamma = 0,
    nafnet_arch.py:162-163) for 240 AdamW iterations on synthetic sigma = 25 denoising pairs (B = 8, 128 x 128; data, init and
    schedule seeded, the HIP gradients are bit-reproducible).  Returns the CPU state dict: the PSNR gates below load it into the HIP
    network and into the oracle -- unscaled, the network's output is an image because the network was trained to produce one."""
    from basicsr.archs import build_network

    torch.manual_seed(0)
    net = build_network(dict(type="NAFNetBaseline", **FULL)).to(dev)
    iters = 240
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.9), weight_decay=0.0, fused=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, iters, eta_min=1e-6)
    first = last = None
    for it in range(iters):
        pairs = [_smooth_pair(5000 + 8 * it + j, size=128) for j in range(8)]
        gt = torch.stack([p[0] for p in pairs]).to(dev)
        lq = torch.stack([p[1] for p in pairs]).to(dev)
        opt.zero_grad(set_to_none=True)
        loss = (net(lq) - gt).abs().mean()
        loss.backward()
        opt.step()
        sched.step()
        last = float(loss)
        first = last if first is None else first
    assert last < 0.35 * first, (first, last)   # it learned to denoise (L1 0.085 -> < 0.03)
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}


def test_psnr_gate_hip_vs_oracle(dev, trained_denoiser):
    """north_star: "PSNR within 0.01 dB".  The released weights / test sets are not available offline; the substitute:
    a NAFNet-64 TRAINED on the synthetic sigma = 25 task (fixture above; no scaling of any layer -- round-2 verdict), 4 held-out
    256 x 256 pairs, PSNR / SSIM of the HIP output and of the oracle's output against the same ground truth, through the uint8
    rounding of metrics/psnr_ssim.py:47-75 as SRModel.nondist_validation applies it."""
    from basicsr.archs import build_network
    from basicsr.metrics import calculate_psnr, calculate_ssim
    from basicsr.models.sr_model import tensor2img_rgb

    sd = trained_denoiser
    net = build_network(dict(type="NAFNetBaseline", **FULL))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    worst_p, worst_s, gains = 0.0, 0.0, []
    f = lambda t: t.clamp(0, 1).numpy()   # noqa: E731  (SRModel.nondist_validation: clamped float BCHW into the metric)
    kw = dict(crop_border=0, test_y_channel=False, image_range=255.0)   # options/all_in_one/test/test_NAFNet_5d.yml
    for i in range(4):
        gt, lq = _smooth_pair(i)
        with torch.no_grad():
            out_h = net(lq[None].to(dev)).cpu()
            out_o, _ = O.nafnet_forward(lq[None], sd)
        h8, o8 = tensor2img_rgb(out_h), tensor2img_rgb(out_o)
        ph, po = calculate_psnr(f(out_h), f(gt[None]), **kw), calculate_psnr(f(out_o), f(gt[None]), **kw)
        sh, so = calculate_ssim(f(out_h), f(gt[None]), **kw), calculate_ssim(f(out_o), f(gt[None]), **kw)
        pin = calculate_psnr(f(lq[None]), f(gt[None]), **kw)
        gains.append(po - pin)
        worst_p, worst_s = max(worst_p, abs(ph - po)), max(worst_s, abs(sh - so))
        assert (h8 != o8).mean() < 2e-3, f"image {i}: {(h8 != o8).mean():.2e} of the uint8 pixels differ"
    print(f"trained NAFNet-64: PSNR gain over the noisy input {min(gains):.2f}..{max(gains):.2f} dB; HIP vs oracle |dPSNR| <= {worst_p:.5f} dB, "
          f"|dSSIM| <= {worst_s:.2e}")
    assert min(gains) > 12.0, gains   # the gate is taken on a network that restores (measured: +18.2..18.4 dB over the noisy input)
    assert worst_p <= 0.01, f"PSNR differs by {worst_p:.4f} dB"
    assert worst_s <= 1e-4, f"SSIM differs by {worst_s:.2e}"


def test_psnr_bf16_storage_vs_fp32(dev, trained_denoiser):
    """How far bf16 STORAGE (act_dtype="bf16", an extension; the reference has no reduced precision) moves the acceptance metric: the
    trained NAFNet-64 and the held-out pairs of the gate above, PSNR of the bf16-storage output against the fp32 HIP output's PSNR.
    Not a parity claim (the <= 0.01 dB gate is the fp32 path's, above) -- a measured bound for the mode's documentation."""
    from basicsr.archs import build_network
    from basicsr.metrics import calculate_psnr

    nets = {}
    for dt in ("fp32", "bf16", "bf16_tail32", "bf16_edge32"):
        net = build_network(dict(type="NAFNetBaseline", act_dtype=dt, **FULL))
        net.load_state_dict(trained_denoiser, strict=True)
        nets[dt] = net.to(dev).eval()
    worst = {"bf16": 0.0, "bf16_tail32": 0.0, "bf16_edge32": 0.0}
    psnr = {dt: [] for dt in nets}
    kw = dict(crop_border=0, test_y_channel=False, image_range=255.0)
    f = lambda t: t.clamp(0, 1).cpu().numpy()   # noqa: E731
    for i in range(8):
        gt, lq = _smooth_pair(i)
        with torch.no_grad():
            outs = {dt: n(lq[None].to(dev)) for dt, n in nets.items()}
        for dt in nets:
            assert outs[dt].dtype == torch.float32
            psnr[dt].append(calculate_psnr(f(outs[dt]), f(gt[None]), **kw))
        for dt in worst:
            worst[dt] = max(worst[dt], abs(psnr["fp32"][-1] - psnr[dt][-1]))
    # what a validation run reports and north_star's 0.01 dB refers to is the MEAN over the set (reference sr_model.py:421-436 sums the
    # per-image metric and divides); single images scatter around it by the training run's last bits (0.0085 ... 0.0105 dB between builds)
    mean = {dt: abs(float(np.mean(psnr["fp32"])) - float(np.mean(psnr[dt]))) for dt in worst}
    print(f"bf16 storage moves the set's PSNR by {mean['bf16']:.4f} dB (single image: at most {worst['bf16']:.4f}); with the last decoder "
          f"group + ending in fp32 by {mean['bf16_tail32']:.4f} dB (single image: at most {worst['bf16_tail32']:.4f})")
    assert worst["bf16"] <= 0.03, f"PSNR differs by {worst['bf16']:.4f} dB"   # measured: 0.0125 dB on the trained network
    # act_dtype = "bf16_tail32" is the bf16 mode that stays inside north_star's 0.01 dB
    assert mean["bf16_tail32"] <= 0.01, f"set PSNR differs by {mean['bf16_tail32']:.4f} dB"
    # single images of this mode are OUTSIDE the per-image gate (LABNOTES.md 2: that is what bf16_edge32 is for): a documentation bound only, and
    # it moves with the last bits of the training run that makes the gate network (0.012 dB before, 0.016 dB after the network-edge convs
    # became MFMA kernels with another fp32 summation order; the mode's own arithmetic did not change)
    assert worst["bf16_tail32"] <= 0.02, f"PSNR of one image differs by {worst['bf16_tail32']:.4f} dB"
    # act_dtype = "bf16_edge32" (everything at full resolution in fp32): EVERY held-out image inside the gate (round-4 verdict, item 6b)
    print(f"bf16_edge32: set PSNR moves by {mean['bf16_edge32']:.4f} dB, single image at most {worst['bf16_edge32']:.4f} dB")
    assert worst["bf16_edge32"] <= 0.01, f"PSNR of one image differs by {worst['bf16_edge32']:.4f} dB"
