"""The opt-in GEMM precision mode "bf16x3" (gemm_x3.hip, include/dcpt_hip.h dcpt_set_gemm_x3): the wide fp32 NT GEMMs on the bf16 matrix
pipe with both operands split into three bfloat16 pieces (x = x0 + x1 + x2 exactly, six piece products, fp32 accumulation).  The mode is
NOT the product default and never the headline; what is established here is that it is fp32-CLASS:

* a level-3 GEMM against an fp64 product: its error is not larger than the exact-fp32-MFMA kernel's;
* the fp32 parity bar unchanged: NAFBlock / NAFNet golden vectors of the real reference at the fp32 tolerances, with the mode FORCED onto
  every eligible launch (min_tiles = 1) -- tools/x3_e2e.sh runs the whole fp32 parity suite this way (52 tests);
* NAFNet-64 [1,1,1,28] on a 256 x 256 image against the oracle evaluated in FLOAT64: output and input-gradient errors of the forced
  bf16x3 run vs the fp32-MFMA run;
* at the bench size (B = 32, where the launcher selects the mode by itself) one training step's loss and gradients against the fp32 step.
"""
import os

import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
from oracle import nafnet_oracle as O

pytestmark = pytest.mark.gpu
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dcpt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture()
def x3(dev):
    from dcpt_amd import functional as DF

    def on(min_tiles=0):
        DF.set_gemm_precision("bf16x3", device=dev, min_tiles=min_tiles)

    yield on
    DF.set_gemm_precision("fp32")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(32, 32, 32, 512, 1024), (32, 32, 32, 1024, 512), (25, 31, 32, 512, 512), (32, 64, 64, 256, 512)])
def test_gemm_error_vs_fp64_not_above_fp32_mfma(dev, x3, shape):
    from dcpt_amd import functional as DF

    B, H, W, Ci, Co = shape
    g = torch.Generator(device=dev).manual_seed(Ci + Co)
    x = torch.randn((B, Ci, H, W), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn((Co, Ci, 1, 1), generator=g, device=dev) / Ci ** 0.5
    M = B * H * W
    ref = x.permute(0, 2, 3, 1).reshape(M, Ci).double() @ w.reshape(Co, Ci).double().t()
    with torch.no_grad():
        y32 = DF.conv_nobias(x, w).permute(0, 2, 3, 1).reshape(M, Co).double()
        x3()
        y3 = DF.conv_nobias(x, w).permute(0, 2, 3, 1).reshape(M, Co).double()
    assert not torch.equal(y32, y3)   # (the mode really ran: the two kernels round differently)
    rms = lambda y: float(((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))   # noqa: E731
    mx = lambda y: float((y - ref).abs().max() / ref.abs().max())   # noqa: E731
    print(f"{shape}: rms error fp32-MFMA {rms(y32):.2e} / bf16x3 {rms(y3):.2e}; max {mx(y32):.2e} / {mx(y3):.2e}")
    assert rms(y3) <= 1.1 * rms(y32) and mx(y3) <= 1.25 * mx(y32), (rms(y3), rms(y32), mx(y3), mx(y32))
    assert rms(y3) <= 1e-6


def test_nafblock_and_tiny_net_goldens_at_fp32_tolerance_with_the_mode_forced(dev, x3, golden_dir):
    """the reference's own vectors (tests/golden/nafblock_c64.npz, nafnet_tiny.npz) at the tolerances of tests/test_gpu_parity.py"""
    import tests.test_gpu_parity as P

    x3(min_tiles=1)
    P.test_nafblock_golden(torch.device("cuda:0"), golden_dir, 64)
    P.test_nafnet_tiny_golden(torch.device("cuda:0"), golden_dir)
    P.test_nafnet_full_golden(torch.device("cuda:0"), golden_dir)


def test_nafnet64_against_the_fp64_oracle(dev, x3):
    """NAFNet-64 [1,1,1,28], one 256 x 256 image, keyed weights: output and input gradient of the fp32-MFMA run and of the forced bf16x3
    run against the oracle evaluated in float64 -- the bf16x3 error must not exceed the fp32-MFMA error by more than 25 %."""
    from basicsr.archs import build_network
    from dcpt_amd import functional as DF

    sd = keyed_state_dict(O.nafnet_param_shapes(**FULL), seed=0)
    x0 = keyed_input("x3.full.x", (1, 3, 256, 256))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    P64 = {k: v.double() for k, v in sd.items()}
    xr = x0.double().requires_grad_(True)
    yr, _ = O.nafnet_forward(xr, P64)
    yr.square().mean().backward()
    net = build_network(dict(type="NAFNetBaseline", **FULL))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    errs = {}
    for mode in ("fp32", "bf16x3"):
        if mode == "bf16x3":
            x3(min_tiles=1)
        x = x0.to(dev).requires_grad_(True)
        y = net(x)
        y.square().mean().backward()
        errs[mode] = (_rel(y, yr), _rel(x.grad, xr.grad))
    print(f"NAFNet-64 vs fp64 oracle: output error fp32-MFMA {errs['fp32'][0]:.2e} / bf16x3 {errs['bf16x3'][0]:.2e}; "
          f"input-gradient error {errs['fp32'][1]:.2e} / {errs['bf16x3'][1]:.2e}")
    assert errs["bf16x3"][0] <= 1.25 * errs["fp32"][0] + 1e-7 and errs["bf16x3"][1] <= 1.25 * errs["fp32"][1] + 1e-7, errs
    assert errs["bf16x3"][0] <= 1e-4


def test_training_step_at_bench_size_matches_fp32(dev, x3):
    """B = 32, 256 x 256: the launcher routes the level-2/3/4 GEMMs to the mode by itself; loss and gradients against the fp32 step."""
    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import fill_module_

    net = fill_module_(build_network(dict(type="NAFNetBaseline", **FULL)), seed=0).to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    lq, gt = torch.rand((32, 3, 256, 256), generator=g, device=dev), torch.rand((32, 3, 256, 256), generator=g, device=dev)
    res = {}
    for mode in ("fp32", "bf16x3"):
        if mode == "bf16x3":
            x3()
        for p in net.parameters():
            p.grad = None
        loss = (net(lq) - gt).abs().mean()
        loss.backward()
        res[mode] = (float(loss), torch.cat([p.grad.detach().double().flatten() for p in net.parameters()]))
    a, b = res["bf16x3"][1], res["fp32"][1]
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print(f"bench-size step: loss fp32 {res['fp32'][0]:.6f} / bf16x3 {res['bf16x3'][0]:.6f}, gradient cosine {cos:.7f}, "
          f"norm ratio {float(a.norm() / b.norm()):.6f}")
    assert not torch.equal(a, b)
    assert abs(res["bf16x3"][0] - res["fp32"][0]) <= 2e-4 * abs(res["fp32"][0])
    assert cos >= 0.9999 and abs(float(a.norm() / b.norm()) - 1.0) <= 1e-3
