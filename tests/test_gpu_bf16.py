"""GPU parity of the bf16-STORAGE NAFBlock (BASELINE.json configs[2]; dcpt_nafblock_fwd_bf16 / bwd_bf16) against the oracle's
bf16 mode (oracle/nafnet_oracle.py::nafblock_bf16: fp32 arithmetic, round-to-nearest-even to bf16 at exactly the points where
the HIP path stores a tensor) and, more loosely, against the fp32 oracle.

Tolerances, stated up front.  The reference has no reduced-precision mode (AMP / TF32 commented out, basicsr/test.py:26-27), so
there is no reference number to match beyond "close to fp32".  bf16 keeps 8 significant bits: one rounding is 2^-9 = 2e-3
relative, and a block stores ~10 tensors in a chain.
  * vs the bf16-mode oracle (same rounding points; what differs is the fp32 summation order inside the MFMA GEMMs, which now
    and then flips a rounding by one bf16 ulp = 4e-3 of that element):  output / input gradient <= 1.5e-2 of the tensor's max,
    parameter gradients <= 2e-2 of their max;
  * vs the fp32 oracle (how far bf16 storage moves the result):  <= 4e-2 / 6e-2 -- a sanity bound, not a parity claim.
"""
import numpy as np
import pytest
import torch

from dcpt_amd.keyed_init import keyed_input, keyed_tensor
from oracle import nafnet_oracle as O

pytestmark = pytest.mark.gpu

FUSED = {"norm1_w": "norm1.weight", "norm1_b": "norm1.bias", "conv1_w": "conv1.weight", "conv1_b": "conv1.bias",
         "conv2_w": "conv2.weight", "conv2_b": "conv2.bias", "conv3_w": "conv3.weight", "conv3_b": "conv3.bias",
         "sca_w": "sca.1.weight", "sca_b": "sca.1.bias", "norm2_w": "norm2.weight", "norm2_b": "norm2.bias",
         "conv4_w": "conv4.weight", "conv4_b": "conv4.bias", "conv5_w": "conv5.weight", "conv5_b": "conv5.bias",
         "beta": "beta", "gamma": "gamma"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from dcpt_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


def _params(c, prefix):
    full = O.nafnet_param_shapes(width=c, enc_blk_nums=[1], middle_blk_num=0, dec_blk_nums=[])
    return {k[len("encoders.0.0."):]: keyed_tensor(prefix + k[len("encoders.0.0."):], s) for k, s in full.items() if k.startswith("encoders.0.0.")}


def _rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    assert a.shape == b.shape
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def test_cast_roundtrip(dev):
    from dcpt_amd import functional as DF

    x = keyed_input("bf.cast", (2, 16, 5, 7), lo=-3, hi=3).to(dev).requires_grad_(True)
    y = DF.to_bf16(x)
    assert y.dtype == torch.bfloat16 and torch.equal(y.float().cpu(), x.detach().cpu().bfloat16().float())   # RNE, as torch
    z = DF.to_f32(y)
    assert z.dtype == torch.float32 and torch.equal(z.cpu(), y.float().cpu())
    z.sum().backward()
    assert x.grad is not None and x.grad.dtype == torch.float32 and float(x.grad.min()) == 1.0


# C = 64 / P = 1024: the fused SCA sums (P % 128 == 0) and image-aligned conv3 weight gradient; (2, 16, 6, 10): ragged tiles,
# C < one MFMA tile; (3, 24, 5, 7): odd image, three images; C = 128 / 512 / 1024: two GEMM column tiles, multi-k-tile loops,
# two LayerNorm chunks per lane
@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (2, 16, 6, 10), (3, 24, 5, 7), (1, 128, 16, 16), (2, 512, 8, 16), (1, 1024, 8, 8)])
def test_nafblock_bf16_oracle(dev, shape):
    from dcpt_amd import functional as DF

    B, c, H, W = shape
    tag = f"bf.{c}.{H}x{W}."
    P = _params(c, tag)
    x = keyed_input(tag + "x", shape, lo=-1.5, hi=1.5).bfloat16().float()
    gw = keyed_input(tag + "gw", shape, lo=-1.0, hi=1.0).bfloat16().float()

    def run_oracle(fn):
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        xr = x.clone().requires_grad_(True)
        y = fn(xr, Pr, "")
        (y * gw).sum().backward()
        return y.detach(), xr.grad, {k: v.grad for k, v in Pr.items()}

    yb, dxb, gb = run_oracle(O.nafblock_bf16)
    yf, dxf, gf = run_oracle(O.nafblock)

    Pd = {k: P[v].to(dev).requires_grad_(True) for k, v in FUSED.items()}
    xd = x.to(dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yd = DF.nafblock_bf16(xd, Pd)
    assert yd.dtype == torch.bfloat16 and yd.shape == xd.shape
    yd.backward(gw.to(dev).bfloat16())
    torch.cuda.synchronize()
    assert xd.grad is not None and xd.grad.dtype == torch.bfloat16
    errs = {"y": _rel(yd, yb), "dx": _rel(xd.grad, dxb)}
    for k, name in FUSED.items():
        errs["d" + k] = _rel(Pd[k].grad, gb[name])
    bad = {k: v for k, v in errs.items() if not np.isfinite(v) or v > (1.5e-2 if k in ("y", "dx") else 2e-2)}
    assert not bad, f"{shape}: vs bf16-mode oracle {bad} (all: { {k: round(v, 4) for k, v in errs.items()} })"
    assert _rel(yd, yf) <= 4e-2 and _rel(xd.grad, dxf) <= 4e-2
    for k, name in FUSED.items():
        assert _rel(Pd[k].grad, gf[name]) <= 6e-2, (k, _rel(Pd[k].grad, gf[name]))


def test_nafnet_bf16_blocks_in_network(dev):
    """``act_dtype='bf16'`` NAFNetBaseline: same state dict, block groups on the bf16 kernels between casts, hooks still fire with
    fp32 group outputs; the network output stays close to the fp32 network's (tiny net, 9 blocks)."""
    from basicsr.archs import build_network
    from dcpt_amd.keyed_init import keyed_state_dict

    cfg = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
    sd = keyed_state_dict(O.nafnet_param_shapes(**cfg), seed=0)
    outs = {}
    for dt in ("fp32", "bf16"):
        net = build_network(dict(type="NAFNetBaseline", act_dtype=dt, **cfg))
        net.load_state_dict(sd, strict=True)
        net = net.to(dev)
        taps = []
        hooks = [getattr(net, f"decoder{i}").register_forward_hook(lambda m, i, o: taps.append(o)) for i in range(4)]
        x = keyed_input("bfnet.x", (2, 3, 32, 32)).to(dev).requires_grad_(True)
        y = net(x)
        y.square().mean().backward()
        assert len(taps) == 4 and all(t.dtype == torch.float32 for t in taps)
        outs[dt] = (y.detach(), x.grad.detach(), {k: p.grad.detach() for k, p in net.named_parameters()})
        for h in hooks:
            h.remove()
    assert _rel(outs["bf16"][0], outs["fp32"][0]) <= 3e-2
    assert _rel(outs["bf16"][1], outs["fp32"][1]) <= 8e-2
    worst = max(_rel(outs["bf16"][2][k], outs["fp32"][2][k]) for k in outs["fp32"][2])
    assert worst <= 0.15, worst


def test_dcpt_step_with_bf16_encoder(dev):
    """DCPTModel.optimize_parameters (reference ...pretrain_model.py:133-169) with ``network_g.act_dtype: bf16``: the decoder taps
    (hooks on ``decoder{i}.0``) still fire with fp32 tensors, the head and both optimizers run, and the losses stay close to the
    fp32 step's."""
    from basicsr.models import build_model
    from dcpt_amd.keyed_init import keyed_state_dict
    from oracle import dc_oracle as D

    tiny = dict(img_channel=3, width=8, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 2], dec_blk_nums=[1, 1, 1, 1])
    dc = dict(feature_dims=[8, 16, 32, 64], num_res_blocks=2, num_classes=10)
    logs = {}
    for dt in ("fp32", "bf16"):
        opt = dict(name="t", model_type="DCPTModel", scale=1, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
                   hook_names="decoder", network_g=dict(type="NAFNetBaseline", act_dtype=dt, **tiny),
                   network_dc=dict(type="PromptIR_NoImg_DC", **dc), path=dict(),
                   train=dict(pixel_opt=dict(type="L1Loss", loss_weight=1.0, reduction="mean"),
                              classify_opt=dict(type="CrossEntropyLoss", loss_weight=1.0),
                              optim_g=dict(type="SGD", lr=0.0), optim_dc=dict(type="SGD", lr=0.0)))
        m = build_model(opt)
        m.net_g.load_state_dict(keyed_state_dict(O.nafnet_param_shapes(**tiny), seed=0), strict=True)
        m.net_dc.load_state_dict(keyed_state_dict(D.dc_param_shapes(**dc), seed=0), strict=True)
        assert len(m.hooks) == 4
        m.feed_data({"lq": keyed_input("dcpt.lq", (2, 3, 32, 32)), "gt": keyed_input("dcpt.gt", (2, 3, 32, 32)),
                     "dataset_idx": torch.tensor([3, 8])})
        m.optimize_parameters(1)
        logs[dt] = dict(m.get_current_log())
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.net_g.parameters())
        assert all(p.grad is not None for p in m.net_dc.parameters()) and m.hook_outputs == []
    assert abs(logs["bf16"]["l_pix"] - logs["fp32"]["l_pix"]) <= 2e-2 * abs(logs["fp32"]["l_pix"]), logs
    assert abs(logs["bf16"]["l_classify"] - logs["fp32"]["l_classify"]) <= 5e-2 * abs(logs["fp32"]["l_classify"]), logs
