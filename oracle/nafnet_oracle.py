"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's
NAFNet hot path in plain PyTorch fp32.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``dcpt_amd`` / ``basicsr``) never does.

Pinned: ``oracle/make_golden.py`` imports the real reference from /root/reference (only in
the build container), loads the same keyed weights into it and stores its outputs and
gradients under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against those vectors.

Each function cites the reference lines it restates.  Everything is functional: the
parameters come in as a flat dict keyed by the reference's ``state_dict`` names, so there
is no nn.Module tree here and nothing shared with the product's module code.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

EPS = 1e-6  # reference nafnet_arch.py:57 (LayerNorm2d default eps)


def layernorm2d(x, weight, bias, eps: float = EPS):
    """Per-pixel LayerNorm over the channel axis of an NCHW tensor.

    reference basicsr/archs/nafnet_arch.py:27-35: mu = mean_c x; var = mean_c (x-mu)^2
    (biased); y = (x-mu)/sqrt(var+eps); out = w[c]*y + b[c].
    """
    mu = x.mean(dim=1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=1, keepdim=True)
    y = xc / torch.sqrt(var + eps)
    return y * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def simple_gate(x):
    """reference nafnet_arch.py:77-80: first half of the channels times the second half."""
    c = x.shape[1] // 2
    return x[:, :c] * x[:, c:]


def nafblock(inp, P: dict, pre: str):
    """reference nafnet_arch.py:165-186 (NAFBlock.forward), dropout rate 0 (Identity)."""
    g = lambda n: P[pre + n]
    c2 = g("conv2.weight").shape[0]
    x = layernorm2d(inp, g("norm1.weight"), g("norm1.bias"))
    x = F.conv2d(x, g("conv1.weight"), g("conv1.bias"))                       # 1x1 c -> 2c
    x = F.conv2d(x, g("conv2.weight"), g("conv2.bias"), padding=1, groups=c2)  # dw 3x3
    x = simple_gate(x)
    pooled = x.mean(dim=(2, 3), keepdim=True)                                  # AdaptiveAvgPool2d(1)
    x = x * F.conv2d(pooled, g("sca.1.weight"), g("sca.1.bias"))               # SCA
    x = F.conv2d(x, g("conv3.weight"), g("conv3.bias"))
    y = inp + x * g("beta")
    x = F.conv2d(layernorm2d(y, g("norm2.weight"), g("norm2.bias")), g("conv4.weight"), g("conv4.bias"))
    x = simple_gate(x)
    x = F.conv2d(x, g("conv5.weight"), g("conv5.bias"))
    return y + x * g("gamma")


# ---- bf16-storage mode (BASELINE.json configs[2]) -----------------------------------------------------------------------
# The reference has no reduced-precision arithmetic (AMP / TF32 are commented out, basicsr/test.py:26-27), so there is nothing
# of the reference to pin this mode to beyond "close to fp32".  It restates nafblock() with a bf16 round-to-nearest-even at every
# point where the HIP bf16 path (dcpt_amd/csrc/nafblock_bf16.hip) stores a tensor to HBM, everything else in fp32.
class _Round(torch.autograd.Function):
    """fwd / bwd flags: round the value in forward and / or the gradient in backward through bf16"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.bfloat16().float() if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.bfloat16().float() if ctx.bwd else g), None, None


def _rr(x):   # stored in bf16 in forward, its gradient is stored in bf16 in backward
    return _Round.apply(x, True, True)


def _rf(x):   # rounded in forward only (weight operand copies; SimpleGate(v))
    return _Round.apply(x, True, False)


def _rb(x):   # only its gradient is stored in bf16
    return _Round.apply(x, False, True)


def nafblock_bf16(inp, P: dict, pre: str):
    """nafblock() with the HIP bf16 path's rounding points.  ``inp`` holds bf16-representable values; so does the result."""
    g = lambda n: P[pre + n]
    c = g("conv3.weight").shape[0]
    c2 = 2 * c
    inp = _rb(inp)                                                                  # dinp is stored in bf16
    xn1 = _rr(layernorm2d(inp, g("norm1.weight"), g("norm1.bias")))
    t1 = _rr(F.conv2d(xn1, _rf(g("conv1.weight")), g("conv1.bias")))
    a = F.conv2d(t1, g("conv2.weight"), g("conv2.bias"), padding=1, groups=c2)     # depthwise: fp32 weights on the bf16 t1
    t2u = simple_gate(a)
    pooled = t2u.mean(dim=(2, 3))                                                    # pooling sums the unrounded gate outputs
    t2 = _rf(t2u)                                                                    # (dt2 = dts * s + dpool stays on chip)
    s = F.linear(pooled, g("sca.1.weight").flatten(1), g("sca.1.bias"))             # [B][C]
    if t2.shape[2] * t2.shape[3] < 2 * c:
        # small images (fewer pixels than 2 x channels): the HIP path scales the ACTIVATIONS, t2 * s stored in bf16 (its gradient too),
        # and runs one GEMM with the shared bf16 weights -- per-image weight copies would be larger than the activations they multiply
        t2s = _rr(t2 * s[:, :, None, None])
        x3 = F.conv2d(t2s, _rf(g("conv3.weight")), g("conv3.bias"))
    else:
        w3s = _rf(g("conv3.weight").flatten(1)[None] * s[:, None, :])               # per-image weights W3[n][k] * s[b][k]
        x3 = torch.einsum("bnk,bkhw->bnhw", w3s, _rb(t2)) + g("conv3.bias").view(1, -1, 1, 1)   # (dts is stored in bf16)
    y = _rr(inp + x3 * g("beta"))
    xn2 = _rr(layernorm2d(y, g("norm2.weight"), g("norm2.bias")))
    vu = F.conv2d(xn2, _rf(g("conv4.weight")), g("conv4.bias"))
    v = _rr(vu)
    gate_val = simple_gate(vu).bfloat16().float()                                    # product of the UNROUNDED halves, rounded once
    gate = gate_val.detach() + (simple_gate(v) - simple_gate(v).detach())            # backward uses the stored (rounded) v
    x5 = F.conv2d(gate, _rf(g("conv5.weight")), g("conv5.bias"))
    return _rr(y + x5 * g("gamma"))


def pixel_shuffle2(x):
    """out[n, k, 2h+i, 2w+j] = in[n, 4k+2i+j, h, w] (torch.nn.PixelShuffle(2))."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.view(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c, 2 * h, 2 * w)


def nafnet_cfg_from_params(P: dict):
    """Recover (enc_blk_nums, middle_blk_num, dec_blk_nums) from state-dict key names."""
    def count(prefix):
        idx = set()
        for k in P:
            if k.startswith(prefix):
                idx.add(int(k[len(prefix):].split(".")[0]))
        return len(idx)

    n_levels = count("downs.")
    enc = [count(f"encoders.{i}.") for i in range(n_levels)]
    mid = count("middle_blks.")
    dec = [count(f"decoder{i}.") for i in range(count("ups."))]
    return enc, mid, dec


def nafnet_forward(inp, P: dict, hook: bool = False):
    """reference nafnet_arch.py:250-274 (NAFNetBaseline.forward).

    Returns (output_or_None, taps) where taps are the decoder{i} group outputs, in the
    order the reference's forward hooks would record them (decoder0 first).
    """
    enc_nums, mid_num, dec_nums = nafnet_cfg_from_params(P)
    x = F.conv2d(inp, P["intro.weight"], P["intro.bias"], padding=1)
    encs = []
    for i, nb in enumerate(enc_nums):
        for j in range(nb):
            x = nafblock(x, P, f"encoders.{i}.{j}.")
        encs.append(x)
        x = F.conv2d(x, P[f"downs.{i}.weight"], P[f"downs.{i}.bias"], stride=2)
    for j in range(mid_num):
        x = nafblock(x, P, f"middle_blks.{j}.")
    taps = []
    for i, nb in enumerate(dec_nums):
        x = pixel_shuffle2(F.conv2d(x, P[f"ups.{i}.0.weight"]))
        x = x + encs[len(encs) - 1 - i]
        for j in range(nb):
            x = nafblock(x, P, f"decoder{i}.{j}.")
        taps.append(x)
    if hook:
        return None, taps
    x = F.conv2d(x, P["ending.weight"], P["ending.bias"], padding=1) + inp
    return x, taps


# ---- bf16-storage mode of the layers between the block groups (dcpt_amd/csrc/edge_bf16.hip) ------------------------------------
# Feature maps and their gradients are stored in bf16 (rounded once, on store), GEMM weight operands are bf16 copies, the two
# 3-channel convs (direct VALU kernels) use the fp32 weights; images, biases, all parameter gradients fp32.
def intro_bf16(inp, w, b):
    return _rr(F.conv2d(inp, w, b, padding=1))


def ending_bf16(x, w, b, inp):
    return F.conv2d(_rb(x), w, b, padding=1) + inp


def down_bf16(x, w, b):
    return _rr(F.conv2d(_rb(x), _rf(w), b, stride=2))


def up_bf16(x, w, skip):
    y = pixel_shuffle2(F.conv2d(_rb(x), _rf(w)))
    return _rr(y + skip if skip is not None else y)


def nafnet_forward_bf16(inp, P: dict, hook: bool = False):
    """nafnet_forward() in bf16 storage end to end (NAFNetBaseline(act_dtype="bf16")): same return convention."""
    enc_nums, mid_num, dec_nums = nafnet_cfg_from_params(P)
    x = intro_bf16(inp, P["intro.weight"], P["intro.bias"])
    encs = []
    for i, nb in enumerate(enc_nums):
        for j in range(nb):
            x = nafblock_bf16(x, P, f"encoders.{i}.{j}.")
        x = _rb(x)   # the group output feeds the down conv AND the skip: autograd adds the two bf16 gradients into a bf16 tensor
        encs.append(x)
        x = down_bf16(x, P[f"downs.{i}.weight"], P[f"downs.{i}.bias"])
    for j in range(mid_num):
        x = nafblock_bf16(x, P, f"middle_blks.{j}.")
    taps = []
    for i, nb in enumerate(dec_nums):
        x = up_bf16(x, P[f"ups.{i}.0.weight"], encs[len(encs) - 1 - i])
        for j in range(nb):
            x = nafblock_bf16(x, P, f"decoder{i}.{j}.")
        taps.append(x)
    if hook:
        return None, taps
    return ending_bf16(x, P["ending.weight"], P["ending.bias"], inp), taps


def nafnet_param_shapes(img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=(), dec_blk_nums=()):
    """State-dict key -> shape, in the reference's registration order
    (nafnet_arch.py:200-248; NAFBlock.__init__ :84-163)."""
    shapes = {}

    def block(pre, c):
        shapes[pre + "beta"] = (1, c, 1, 1)
        shapes[pre + "gamma"] = (1, c, 1, 1)
        shapes[pre + "conv1.weight"] = (2 * c, c, 1, 1)
        shapes[pre + "conv1.bias"] = (2 * c,)
        shapes[pre + "conv2.weight"] = (2 * c, 1, 3, 3)
        shapes[pre + "conv2.bias"] = (2 * c,)
        shapes[pre + "conv3.weight"] = (c, c, 1, 1)
        shapes[pre + "conv3.bias"] = (c,)
        shapes[pre + "sca.1.weight"] = (c, c, 1, 1)
        shapes[pre + "sca.1.bias"] = (c,)
        shapes[pre + "conv4.weight"] = (2 * c, c, 1, 1)
        shapes[pre + "conv4.bias"] = (2 * c,)
        shapes[pre + "conv5.weight"] = (c, c, 1, 1)
        shapes[pre + "conv5.bias"] = (c,)
        shapes[pre + "norm1.weight"] = (c,)
        shapes[pre + "norm1.bias"] = (c,)
        shapes[pre + "norm2.weight"] = (c,)
        shapes[pre + "norm2.bias"] = (c,)

    shapes["intro.weight"] = (width, img_channel, 3, 3)
    shapes["intro.bias"] = (width,)
    shapes["ending.weight"] = (img_channel, width, 3, 3)
    shapes["ending.bias"] = (img_channel,)
    chan = width
    for i, num in enumerate(enc_blk_nums):
        for j in range(num):
            block(f"encoders.{i}.{j}.", chan)
        chan *= 2
    for j in range(middle_blk_num):
        block(f"middle_blks.{j}.", chan)
    c = chan
    for i, num in enumerate(dec_blk_nums):
        shapes[f"ups.{i}.0.weight"] = (2 * c, c, 1, 1)
        c //= 2
    c = width
    for i, _ in enumerate(enc_blk_nums):
        shapes[f"downs.{i}.weight"] = (2 * c, c, 2, 2)
        shapes[f"downs.{i}.bias"] = (2 * c,)
        c *= 2
    c = chan
    for i, num in enumerate(dec_blk_nums):
        c //= 2
        for j in range(num):
            block(f"decoder{i}.{j}.", c)
    return shapes


def tlsc_avgpool(x, kernel_size):
    """TLSC local average pool used by the ``NAFNet`` (Local_Base) variant.

    reference basicsr/archs/arch_util.py:352-396 (non-fast path): global mean when the
    kernel covers the image, else a k1 x k2 box mean from a 2-D prefix sum, replicate-padded
    back to H x W.
    """
    n, c, h, w = x.shape
    k1, k2 = kernel_size
    if k1 >= h and k2 >= w:
        return x.mean(dim=(2, 3), keepdim=True)
    s = x.cumsum(dim=-1).cumsum(dim=-2)
    s = F.pad(s, (1, 0, 1, 0))
    k1, k2 = min(h, k1), min(w, k2)
    out = (s[:, :, k1:, k2:] + s[:, :, :-k1, :-k2] - s[:, :, :-k1, k2:] - s[:, :, k1:, :-k2]) / (k1 * k2)
    _h, _w = out.shape[2:]
    pad = ((w - _w) // 2, (w - _w + 1) // 2, (h - _h) // 2, (h - _h + 1) // 2)
    return F.pad(out, pad, mode="replicate")


def l1_loss(a, b):
    """reference basicsr/losses/basic_loss.py L1Loss with reduction='mean'."""
    return (a - b).abs().mean()


def psnr_uint8(a, b):
    """reference basicsr/metrics/psnr_ssim.py:47-75 as used by the YAMLs: clamp to [0,1],
    x255, round to uint8, full RGB, crop_border 0, float64 mse, 10*log10(255^2/mse)."""
    import numpy as np

    a8 = (a.detach().clamp(0, 1) * 255.0).round().to(torch.uint8).numpy().astype(np.float64)
    b8 = (b.detach().clamp(0, 1) * 255.0).round().to(torch.uint8).numpy().astype(np.float64)
    mse = ((a8 - b8) ** 2).mean()
    if mse == 0:
        return float("inf")
    return float(10.0 * np.log10(255.0 * 255.0 / mse))
