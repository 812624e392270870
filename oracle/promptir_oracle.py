"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's PromptIR
(basicsr/archs/promptir_arch.py) in plain PyTorch fp32, functional over a flat state-dict-keyed parameter dict.
Pinned by tests/golden/promptir_*.npz (oracle/make_golden.py imports the real reference)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .restormer_oracle import feedforward


def layernorm(x, P, pre):
    """promptir_arch.py:27-59 on the (b, hw, c) view: eps 1e-5 (Restormer's copy in this repo uses 1e-6); BiasFree when there
    is no bias key (variance about the mean, no mean subtraction in the numerator, :39-40)."""
    b, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    sigma = t.var(-1, keepdim=True, unbiased=False)
    if pre + "body.bias" in P:
        t = (t - t.mean(-1, keepdim=True)) / torch.sqrt(sigma + 1e-5) * P[pre + "body.weight"] + P[pre + "body.bias"]
    else:
        t = t / torch.sqrt(sigma + 1e-5) * P[pre + "body.weight"]
    return t.reshape(b, h, w, c).permute(0, 3, 1, 2)


def attention(x, P, pre):
    """MDTA, promptir_arch.py:115-145: softmax over the last dimension (:136)."""
    b, c, h, w = x.shape
    heads = P[pre + "temperature"].shape[0]
    qkv = F.conv2d(F.conv2d(x, P[pre + "qkv.weight"]), P[pre + "qkv_dwconv.weight"], padding=1, groups=3 * c)
    q, k, v = qkv.chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, c // heads, h * w) for t in (q, k, v))
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    attn = ((q @ k.transpose(-2, -1)) * P[pre + "temperature"]).softmax(dim=-1)
    out = (attn @ v).reshape(b, c, h, w)
    return F.conv2d(out, P[pre + "project_out.weight"])


def transformer_block(x, P, pre):
    x = x + attention(layernorm(x, P, pre + "norm1."), P, pre + "attn.")
    return x + feedforward(layernorm(x, P, pre + "norm2."), P, pre + "ffn.")   # GDFN is identical to Restormer's


def _level(x, P, pre):
    idx = sorted({int(k[len(pre):].split(".")[0]) for k in P if k.startswith(pre)})
    for i in idx:
        x = transformer_block(x, P, f"{pre}{i}.")
    return x


def prompt_block(x, P, pre):
    """PromptGenBlock.forward (:249-262)"""
    B, C, H, W = x.shape
    emb = x.mean(dim=(-2, -1))
    w = F.softmax(F.linear(emb, P[pre + "linear_layer.weight"], P[pre + "linear_layer.bias"]), dim=1)
    prompt = (w[:, :, None, None, None] * P[pre + "prompt_param"]).sum(dim=1)   # (B, L, 1,1,1) * (1, L, D, S, S)
    prompt = F.interpolate(prompt, (H, W), mode="bilinear")
    return F.conv2d(prompt, P[pre + "conv3x3.weight"], padding=1)


def _prompted(x, P, prompt, noise, reduce):
    x = torch.cat([x, prompt_block(x, P, prompt + ".")], 1)
    return F.conv2d(transformer_block(x, P, noise + "."), P[reduce + ".weight"])


def promptir_forward(inp, P, hook=False):
    """PromptIR.forward (:468-518) with decoder=True; hook=True returns None after the first prompt stage of the decoder."""
    e1 = _level(F.conv2d(inp, P["patch_embed.proj.weight"], padding=1), P, "encoder_level1.")
    e2 = _level(F.pixel_unshuffle(F.conv2d(e1, P["down1_2.body.0.weight"], padding=1), 2), P, "encoder_level2.")
    e3 = _level(F.pixel_unshuffle(F.conv2d(e2, P["down2_3.body.0.weight"], padding=1), 2), P, "encoder_level3.")
    lat = _level(F.pixel_unshuffle(F.conv2d(e3, P["down3_4.body.0.weight"], padding=1), 2), P, "latent.")
    lat = _prompted(lat, P, "prompt3", "noise_level3", "reduce_noise_level3")
    d3 = torch.cat([F.pixel_shuffle(F.conv2d(lat, P["up4_3.body.0.weight"], padding=1), 2), e3], 1)
    d3 = _level(F.conv2d(d3, P["reduce_chan_level3.weight"]), P, "decoder_level3.")
    d3 = _prompted(d3, P, "prompt2", "noise_level2", "reduce_noise_level2")
    d2 = torch.cat([F.pixel_shuffle(F.conv2d(d3, P["up3_2.body.0.weight"], padding=1), 2), e2], 1)
    d2 = _level(F.conv2d(d2, P["reduce_chan_level2.weight"]), P, "decoder_level2.")
    d2 = _prompted(d2, P, "prompt1", "noise_level1", "reduce_noise_level1")
    if hook:
        return None
    d1 = torch.cat([F.pixel_shuffle(F.conv2d(d2, P["up2_1.body.0.weight"], padding=1), 2), e1], 1)
    d1 = _level(_level(d1, P, "decoder_level1."), P, "refinement.")
    return F.conv2d(d1, P["output.weight"], padding=1) + inp
