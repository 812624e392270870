"""ORACLE (test infrastructure, not product code) -- CPU restatement of the degradation-classifier head
``PromptIR_NoImg_DC`` (reference basicsr/archs/degrad_classify_arch.py:558-641) and of ``PromptIR_DC`` (:480-555, the
variant seeded by a 7x7 stride-2 embedding of the image) in plain PyTorch fp32.
Pinned by tests/golden/dc_head.npz and tests/golden/dc_img_head.npz (oracle/make_golden.py imports the real reference).
Functional: parameters come as a flat dict keyed by the reference's state-dict names."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def layernorm_cf(x, w, b, eps=1e-6):
    """channels_first LayerNorm (degrad_classify_arch.py:39-44)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def bottleneck(x, P, pre):
    """BottleneckBlock.forward (:227-243) with in==out channels (identity shortcut), norm=LN, no conv bias."""
    g = lambda n: P[pre + n]
    out = F.relu(layernorm_cf(F.conv2d(x, g("conv1.weight")), g("conv1.norm.weight"), g("conv1.norm.bias")))
    out = F.relu(layernorm_cf(F.conv2d(out, g("conv2.weight"), padding=1), g("conv2.norm.weight"), g("conv2.norm.bias")))
    out = layernorm_cf(F.conv2d(out, g("conv3.weight")), g("conv3.norm.weight"), g("conv3.norm.bias"))
    return F.relu(out + x)


def dc_forward(features, P, x0=0):
    """PromptIR_NoImg_DC.forward (:621-641) with downsample=False; ``lq`` is ignored by the reference.
    features: list of NCHW maps, highest resolution first.  ``x0``: running map before the first stage."""
    n = len(features)
    mix = torch.softmax(P["mixing_weights"], dim=0)  # F.softmax without dim on a 1-D tensor -> dim 0 (:633)
    nblk = len({k.split(".")[2] for k in P if k.startswith("bottleneck_layers.0.")})
    x = x0
    for i, f in enumerate(features):
        x = x + mix[i] * f
        for b in range(nblk):
            x = bottleneck(x, P, f"bottleneck_layers.{i}.{b}.")
        x = F.relu(F.max_pool2d(F.conv2d(x, P[f"downsample_layers.{i}.0.weight"]), 2, 2))
    for b in range(len({k.split(".")[1] for k in P if k.startswith("last_stage.")})):
        x = bottleneck(x, P, f"last_stage.{b}.")
    x = x.mean(dim=[-1, -2])
    return F.linear(x, P["fc.weight"], P["fc.bias"])


# ---- bf16-storage mode of the head (dcpt_amd/csrc/dchead_bf16.hip): fp32 arithmetic, a bf16 round-to-nearest-even wherever the HIP
# path stores an activation (conv output z, LayerNorm-group output y, the stage input after mixing) or its gradient; weights are
# rounded as GEMM operands only.  The reference has no such mode (basicsr/test.py:26-27); pinned to the fp32 head by tolerance.
def _round_ops():
    from oracle.nafnet_oracle import _rf, _rr

    return _rr, _rf


def bottleneck_bf16(x, P, pre):
    rr, rf = _round_ops()
    g = lambda n: P[pre + n]
    z = rr(F.conv2d(x, rf(g("conv1.weight"))))
    out = rr(F.relu(layernorm_cf(z, g("conv1.norm.weight"), g("conv1.norm.bias"))))
    z = rr(F.conv2d(out, rf(g("conv2.weight")), padding=1))
    out = rr(F.relu(layernorm_cf(z, g("conv2.norm.weight"), g("conv2.norm.bias"))))
    z = rr(F.conv2d(out, rf(g("conv3.weight"))))
    return rr(F.relu(layernorm_cf(z, g("conv3.norm.weight"), g("conv3.norm.bias")) + x))


def dc_forward_bf16(features, P):
    rr, rf = _round_ops()
    mix = torch.softmax(P["mixing_weights"], dim=0)
    nblk = len({k.split(".")[2] for k in P if k.startswith("bottleneck_layers.0.")})
    x = 0
    for i, f in enumerate(features):
        x = rr(x + mix[i] * f)                     # the mixing runs in fp32, its result enters the stage as bf16
        for b in range(nblk):
            x = bottleneck_bf16(x, P, f"bottleneck_layers.{i}.{b}.")
        z = rr(F.conv2d(x, rf(P[f"downsample_layers.{i}.0.weight"])))
        x = rr(F.relu(F.max_pool2d(z, 2, 2)))
    for b in range(len({k.split(".")[1] for k in P if k.startswith("last_stage.")})):
        x = bottleneck_bf16(x, P, f"last_stage.{b}.")
    x = x.mean(dim=[-1, -2])
    return F.linear(x, P["fc.weight"], P["fc.bias"])


def dc_img_forward(lq, features, P):
    """PromptIR_DC.forward (:546-555): lq_feats = LayerNorm(Conv2d(3, dim0, 7, stride 2, pad 3)(lq)) (:491-494), then the
    same stages as above starting from lq_feats (features[0] lives at half the image resolution)."""
    e = F.conv2d(lq, P["conv_embed.0.weight"], P["conv_embed.0.bias"], stride=2, padding=3)
    e = layernorm_cf(e, P["conv_embed.1.weight"], P["conv_embed.1.bias"])
    return dc_forward(features, P, x0=e)


def dc_param_shapes(feature_dims, num_res_blocks=2, num_classes=3, img_embed=False):
    shapes = {"mixing_weights": (len(feature_dims),)}
    if img_embed:
        shapes["conv_embed.0.weight"] = (feature_dims[0], 3, 7, 7)
        shapes["conv_embed.0.bias"] = (feature_dims[0],)
        shapes["conv_embed.1.weight"] = (feature_dims[0],)
        shapes["conv_embed.1.bias"] = (feature_dims[0],)

    def block(pre, c):
        for name, (co, ci, k) in (("conv1", (2 * c, c, 1)), ("conv2", (2 * c, 2 * c, 3)), ("conv3", (c, 2 * c, 1))):
            shapes[f"{pre}{name}.weight"] = (co, ci, k, k)
            shapes[f"{pre}{name}.norm.weight"] = (co,)
            shapes[f"{pre}{name}.norm.bias"] = (co,)

    for l, c in enumerate(feature_dims):
        for b in range(num_res_blocks):
            block(f"bottleneck_layers.{l}.{b}.", c)
    for l, c in enumerate(feature_dims):
        nxt = feature_dims[l + 1] if l < len(feature_dims) - 1 else c
        shapes[f"downsample_layers.{l}.0.weight"] = (nxt, c, 1, 1)
    for b in range(num_res_blocks):
        block(f"last_stage.{b}.", feature_dims[-1])
    shapes["fc.weight"] = (num_classes, feature_dims[-1])
    shapes["fc.bias"] = (num_classes,)
    return shapes
