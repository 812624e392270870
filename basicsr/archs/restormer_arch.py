"""Restormer for DCPT, MI355X-native.

Same registry names (``Restormer``, ``Restormer_origin``), constructor kwargs, ``forward(inp_img, hook=None)`` contract,
module tree and ``state_dict`` keys/shapes as the reference (basicsr/archs/restormer_arch.py:234-517).  Each
TransformerBlock is two autograd nodes: ``dcpt_mdta_*`` (LayerNorm -> qkv 1x1 -> depthwise 3x3 -> transposed ReLU channel
attention as batched MFMA GEMMs -> project_out + residual) and ``dcpt_gdfn_*`` (LayerNorm -> project_in -> depthwise 3x3 ->
GELU gate -> project_out + residual).  Down/Upsample are a dense 3x3 implicit-GEMM conv + an NHWC pixel (un)shuffle; skip
connections are NHWC channel concats.  Child modules only own the parameters.
"""
from __future__ import annotations

import numbers

import torch
import torch.nn as nn

from basicsr.utils.registry import ARCH_REGISTRY
from dcpt_amd import functional as DF


class BiasFree_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        self.weight = nn.Parameter(torch.ones(torch.Size(normalized_shape)))
        self.normalized_shape = torch.Size(normalized_shape)


class WithBias_LayerNorm(nn.Module):
    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        self.weight = nn.Parameter(torch.ones(torch.Size(normalized_shape)))
        self.bias = nn.Parameter(torch.zeros(torch.Size(normalized_shape)))
        self.normalized_shape = torch.Size(normalized_shape)


class LayerNorm(nn.Module):
    """parameter holder (reference :62-72); the normalisation runs inside the fused MDTA / GDFN ops."""

    def __init__(self, dim, LayerNorm_type):
        super().__init__()
        self.biasfree = LayerNorm_type == "BiasFree"
        self.body = BiasFree_LayerNorm(dim) if self.biasfree else WithBias_LayerNorm(dim)

    def wb(self):
        return self.body.weight, (None if self.biasfree else self.body.bias)


class FeedForward(nn.Module):
    def __init__(self, dim, ffn_expansion_factor, bias):
        super().__init__()
        hidden = int(dim * ffn_expansion_factor)
        self.project_in = nn.Conv2d(dim, hidden * 2, kernel_size=1, bias=False)
        self.dwconv = nn.Conv2d(hidden * 2, hidden * 2, kernel_size=3, stride=1, padding=1, groups=hidden * 2, bias=False)
        self.project_out = nn.Conv2d(hidden, dim, kernel_size=1, bias=False)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, bias):
        super().__init__()
        self.num_heads = num_heads
        self.temperature = nn.Parameter(torch.ones(num_heads, 1, 1))
        self.qkv = nn.Conv2d(dim, dim * 3, kernel_size=1, bias=False)
        self.qkv_dwconv = nn.Conv2d(dim * 3, dim * 3, kernel_size=3, stride=1, padding=1, groups=dim * 3, bias=False)
        self.project_out = nn.Conv2d(dim, dim, kernel_size=1, bias=False)


class TransformerBlock(nn.Module):
    eps_1e5 = False   # PromptIR's copies of these blocks use LayerNorm eps 1e-5 ...
    softmax = False   # ... and softmax instead of ReLU attention (basicsr/archs/promptir_arch.py)

    def __init__(self, dim, num_heads, ffn_expansion_factor, bias, LayerNorm_type):
        super().__init__()
        if bias:
            raise NotImplementedError("the reference hard-codes bias=False in Attention/FeedForward (restormer_arch.py:83,109)")
        if dim % num_heads or (dim // num_heads) % 4:
            raise ValueError(f"dim={dim} / heads={num_heads} must be a multiple of 4 for the MI355X kernels")
        self.norm1 = LayerNorm(dim, LayerNorm_type)
        self.attn = Attention(dim, num_heads, bias)
        self.norm2 = LayerNorm(dim, LayerNorm_type)
        self.ffn = FeedForward(dim, ffn_expansion_factor, bias)

    def forward(self, x):
        w1, b1 = self.norm1.wb()
        x = DF.mdta(x, w1, b1, self.attn.qkv.weight, self.attn.qkv_dwconv.weight, self.attn.project_out.weight,
                    self.attn.temperature, self.attn.num_heads, self.norm1.biasfree, self.eps_1e5, self.softmax)
        w2, b2 = self.norm2.wb()
        return DF.gdfn(x, w2, b2, self.ffn.project_in.weight, self.ffn.dwconv.weight, self.ffn.project_out.weight,
                       self.norm2.biasfree, self.eps_1e5)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, in_c=3, embed_dim=48, bias=False):
        super().__init__()
        self.proj = nn.Conv2d(in_c, embed_dim, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return DF.conv3x3_in(x, self.proj.weight, None)


class Downsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat // 2, kernel_size=3, stride=1, padding=1, bias=False), nn.PixelUnshuffle(2))

    def forward(self, x):
        return DF.pixel_unshuffle2(DF.conv_nobias(x, self.body[0].weight))


class Upsample(nn.Module):
    def __init__(self, n_feat):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(n_feat, n_feat * 2, kernel_size=3, stride=1, padding=1, bias=False), nn.PixelShuffle(2))

    def forward(self, x):
        return DF.pixel_shuffle2(DF.conv_nobias(x, self.body[0].weight))


class SequentialTransformerBlock(nn.Module):
    def __init__(self, dim, head, num_block, ffn_expansion_factor=2.66, bias=False, LayerNorm_type="BiasFree"):
        super().__init__()
        self.body = nn.Sequential(*[TransformerBlock(dim, head, ffn_expansion_factor, bias, LayerNorm_type) for _ in range(num_block)])

    def forward(self, x):
        return self.body(x)


def _trunc_normal_(t, std=0.02):
    nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class _RestormerBase(nn.Module):
    def _set_save_mode(self, save_mode):
        # ``network_g.save_mode`` (this repo's extension): what the transformer blocks keep for backward -- "full" | "balanced" | "lean" |
        # "auto" (dcpt_amd/functional.py, "What the Restormer halves keep"); None = the process default ("balanced").  Results are
        # bit-identical in every mode; only step time and peak memory differ.
        if save_mode not in (None, "auto", "full", "balanced", "lean"):
            raise ValueError(f"save_mode must be 'full', 'balanced', 'lean' or 'auto', got {save_mode!r}")
        self.save_mode = save_mode

    def _build(self, inp_channels, out_channels, dim, num_blocks, num_refinement_blocks, heads, ffn_expansion_factor, bias,
               LayerNorm_type, make_level):
        if bias:
            raise NotImplementedError("bias=True convs are not on the DCPT path")
        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.encoder_level1 = make_level(dim, heads[0], num_blocks[0])
        self.down1_2 = Downsample(dim)
        self.encoder_level2 = make_level(dim * 2, heads[1], num_blocks[1])
        self.down2_3 = Downsample(dim * 2)
        self.encoder_level3 = make_level(dim * 4, heads[2], num_blocks[2])
        self.down3_4 = Downsample(dim * 4)
        self.latent = make_level(dim * 8, heads[3], num_blocks[3])
        self.up4_3 = Upsample(dim * 8)
        self.reduce_chan_level3 = nn.Conv2d(dim * 8, dim * 4, kernel_size=1, bias=False)
        self.decoder_level3 = make_level(dim * 4, heads[2], num_blocks[2])
        self.up3_2 = Upsample(dim * 4)
        self.reduce_chan_level2 = nn.Conv2d(dim * 4, dim * 2, kernel_size=1, bias=False)
        self.decoder_level2 = make_level(dim * 2, heads[1], num_blocks[1])
        self.up2_1 = Upsample(dim * 2)
        self.decoder_level1 = make_level(dim * 2, heads[0], num_blocks[0])
        self.refinement = make_level(dim * 2, heads[0], num_refinement_blocks)
        self.output = nn.Conv2d(dim * 2, out_channels, kernel_size=3, stride=1, padding=1, bias=False)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _trunc_normal_(m.weight, std=0.02)

    def _features(self, inp_img):
        e1 = self.encoder_level1(self.patch_embed(inp_img))
        e2 = self.encoder_level2(self.down1_2(e1))
        e3 = self.encoder_level3(self.down2_3(e2))
        lat = self.latent(self.down3_4(e3))
        d3 = DF.conv_nobias(DF.concat_channels(self.up4_3(lat), e3), self.reduce_chan_level3.weight)
        d3 = self.decoder_level3(d3)
        d2 = DF.conv_nobias(DF.concat_channels(self.up3_2(d3), e2), self.reduce_chan_level2.weight)
        d2 = self.decoder_level2(d2)
        d1 = self.decoder_level1(DF.concat_channels(self.up2_1(d2), e1))
        return d1

    def _tail(self, d1, inp_img):
        return DF.conv3x3_out(self.refinement(d1), self.output.weight, None, inp_img)


@ARCH_REGISTRY.register()
class Restormer(_RestormerBase):
    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type="BiasFree", dual_pixel_task=False,
                 scale=1, window_size=8, save_mode=None):
        super().__init__()
        self._set_save_mode(save_mode)
        if dual_pixel_task or scale != 1:
            raise NotImplementedError("dual_pixel_task / scale > 1 are not on the DCPT path")
        self.dual_pixel_task, self.scale = dual_pixel_task, scale
        self._build(inp_channels, out_channels, dim, num_blocks, num_refinement_blocks, heads, ffn_expansion_factor, bias,
                    LayerNorm_type,
                    lambda d, h, n: SequentialTransformerBlock(d, h, n, ffn_expansion_factor, bias, LayerNorm_type))

    def forward(self, inp_img, hook=None):
        with DF.restormer_save(self.save_mode, inp_img.device):
            d1 = self._features(inp_img)
            if not hook:
                return self._tail(d1, inp_img)
        return None


@ARCH_REGISTRY.register()
class Restormer_origin(_RestormerBase):
    """reference :425-517: plain nn.Sequential levels, WithBias LayerNorm default, no ``hook`` argument."""

    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias", dual_pixel_task=False, save_mode=None):
        super().__init__()
        self._set_save_mode(save_mode)
        if dual_pixel_task:
            raise NotImplementedError("dual_pixel_task is not on the DCPT path")
        self.dual_pixel_task = dual_pixel_task
        self._build(inp_channels, out_channels, dim, num_blocks, num_refinement_blocks, heads, ffn_expansion_factor, bias,
                    LayerNorm_type,
                    lambda d, h, n: nn.Sequential(*[TransformerBlock(d, h, ffn_expansion_factor, bias, LayerNorm_type) for _ in range(n)]))

    def forward(self, inp_img):
        with DF.restormer_save(self.save_mode, inp_img.device):
            return self._tail(self._features(inp_img), inp_img)
