"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's Restormer
(basicsr/archs/restormer_arch.py) in plain PyTorch fp32, functional over a flat state-dict-keyed parameter dict.
Pinned by tests/golden/restormer_*.npz (oracle/make_golden.py imports the real reference)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def layernorm(x, P, pre):
    """restormer_arch.py:26-72 on the (b, hw, c) view; BiasFree when there is no bias key: x/sqrt(var+1e-6)*w with the
    variance about the mean but NO mean subtraction in the numerator (:39-40)."""
    b, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    sigma = t.var(-1, keepdim=True, unbiased=False)
    if pre + "body.bias" in P:
        t = (t - t.mean(-1, keepdim=True)) / torch.sqrt(sigma + 1e-6) * P[pre + "body.weight"] + P[pre + "body.bias"]
    else:
        t = t / torch.sqrt(sigma + 1e-6) * P[pre + "body.weight"]
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2)


def attention(x, P, pre):
    """MDTA, restormer_arch.py:121-145: ReLU (not softmax) channel attention."""
    b, c, h, w = x.shape
    heads = P[pre + "temperature"].shape[0]
    qkv = F.conv2d(F.conv2d(x, P[pre + "qkv.weight"]), P[pre + "qkv_dwconv.weight"], padding=1, groups=3 * c)
    q, k, v = qkv.chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, c // heads, h * w) for t in (q, k, v))
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = F.relu((q @ k.transpose(-2, -1)) * P[pre + "temperature"])
    out = (attn @ v).reshape(b, c, h, w)
    return F.conv2d(out, P[pre + "project_out.weight"])


def feedforward(x, P, pre):
    """GDFN, restormer_arch.py:95-100 (exact erf GELU)."""
    c2 = P[pre + "dwconv.weight"].shape[0]
    x = F.conv2d(F.conv2d(x, P[pre + "project_in.weight"]), P[pre + "dwconv.weight"], padding=1, groups=c2)
    x1, x2 = x.chunk(2, dim=1)
    return F.conv2d(F.gelu(x1) * x2, P[pre + "project_out.weight"])


def transformer_block(x, P, pre):
    x = x + attention(layernorm(x, P, pre + "norm1."), P, pre + "attn.")
    return x + feedforward(layernorm(x, P, pre + "norm2."), P, pre + "ffn.")


def _level(x, P, pre):
    idx = sorted({int(k[len(pre):].split(".")[0]) for k in P if k.startswith(pre)})
    for i in idx:
        x = transformer_block(x, P, f"{pre}{i}.")
    return x


def restormer_forward(inp, P, hook=False, origin=False):
    """Restormer.forward (:376-422) / Restormer_origin.forward; returns (output_or_None, [dec3, dec2, dec1] taps)."""
    lv = (lambda n: n + ".") if origin else (lambda n: n + ".body.")
    x1 = _level(F.conv2d(inp, P["patch_embed.proj.weight"], padding=1), P, lv("encoder_level1"))
    x2 = _level(F.pixel_unshuffle(F.conv2d(x1, P["down1_2.body.0.weight"], padding=1), 2), P, lv("encoder_level2"))
    x3 = _level(F.pixel_unshuffle(F.conv2d(x2, P["down2_3.body.0.weight"], padding=1), 2), P, lv("encoder_level3"))
    lat = _level(F.pixel_unshuffle(F.conv2d(x3, P["down3_4.body.0.weight"], padding=1), 2), P, lv("latent"))
    d3 = torch.cat([F.pixel_shuffle(F.conv2d(lat, P["up4_3.body.0.weight"], padding=1), 2), x3], 1)
    d3 = _level(F.conv2d(d3, P["reduce_chan_level3.weight"]), P, lv("decoder_level3"))
    d2 = torch.cat([F.pixel_shuffle(F.conv2d(d3, P["up3_2.body.0.weight"], padding=1), 2), x2], 1)
    d2 = _level(F.conv2d(d2, P["reduce_chan_level2.weight"]), P, lv("decoder_level2"))
    d1 = torch.cat([F.pixel_shuffle(F.conv2d(d2, P["up2_1.body.0.weight"], padding=1), 2), x1], 1)
    d1 = _level(d1, P, lv("decoder_level1"))
    taps = [d3, d2, d1]
    if hook:
        return None, taps
    out = _level(d1, P, lv("refinement"))
    return F.conv2d(out, P["output.weight"], padding=1) + inp, taps
