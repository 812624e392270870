"""PromptIR for DCPT, MI355X-native.

Same registry name, constructor kwargs, ``forward(inp_img, hook=False)`` contract, module tree and ``state_dict`` keys/shapes
as the reference (basicsr/archs/promptir_arch.py:266-518).  The transformer blocks are Restormer's MDTA / GDFN kernels in
their PromptIR variants (LayerNorm eps 1e-5, softmax attention: ``DCPT_LN_EPS_1E5 | DCPT_ATTN_SOFTMAX``); a prompt block is
``dcpt_meanpool_fc`` (mean over the pixels + linear) -> ``dcpt_prompt_mix`` (softmax-weighted sum of the prompt components,
bilinear resize, NHWC) -> a dense 3x3 implicit-GEMM conv; the prompt is concatenated in NHWC, refined by one block
(``noise_level*``) and reduced by a 1x1 GEMM.  Like the reference, the prompt / channel sizes are hard-wired for ``dim=48``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from basicsr.utils.registry import ARCH_REGISTRY
from dcpt_amd import functional as DF

from .restormer_arch import Downsample, OverlapPatchEmbed, Upsample
from .restormer_arch import TransformerBlock as _RestormerBlock


class TransformerBlock(_RestormerBlock):
    """reference :202-216 with LayerNorm eps 1e-5 (:39-40, :57-59) and softmax attention (:136)"""
    eps_1e5 = True
    softmax = True


class PromptGenBlock(nn.Module):
    """reference :237-262"""

    def __init__(self, prompt_dim=128, prompt_len=5, prompt_size=96, lin_dim=192):
        super().__init__()
        self.prompt_param = nn.Parameter(torch.rand(1, prompt_len, prompt_dim, prompt_size, prompt_size), requires_grad=True)
        self.linear_layer = nn.Linear(lin_dim, prompt_len)
        self.conv3x3 = nn.Conv2d(prompt_dim, prompt_dim, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, x):
        _, _, H, W = x.shape
        logits = DF.meanpool_fc(x, self.linear_layer.weight, self.linear_layer.bias)
        return DF.conv_nobias(DF.prompt_mix(logits, self.prompt_param, H, W), self.conv3x3.weight)


@ARCH_REGISTRY.register()
class PromptIR(nn.Module):
    def __init__(self, inp_channels=3, out_channels=3, dim=48, num_blocks=[4, 6, 6, 8], num_refinement_blocks=4,
                 heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False, LayerNorm_type="WithBias", decoder=True,
                 window_size=8):
        super().__init__()
        if bias:
            raise NotImplementedError("bias=True convs are not on the DCPT path (the reference's option files keep bias=False)")

        def blocks(d, h, n):
            return nn.Sequential(*[TransformerBlock(d, h, ffn_expansion_factor, bias, LayerNorm_type) for _ in range(n)])

        self.patch_embed = OverlapPatchEmbed(inp_channels, dim)
        self.decoder = decoder
        if self.decoder:
            self.prompt1 = PromptGenBlock(prompt_dim=64, prompt_len=5, prompt_size=64, lin_dim=96)
            self.prompt2 = PromptGenBlock(prompt_dim=128, prompt_len=5, prompt_size=32, lin_dim=192)
            self.prompt3 = PromptGenBlock(prompt_dim=320, prompt_len=5, prompt_size=16, lin_dim=384)
        self.encoder_level1 = blocks(dim, heads[0], num_blocks[0])
        self.down1_2 = Downsample(dim)
        self.encoder_level2 = blocks(int(dim * 2 ** 1), heads[1], num_blocks[1])
        self.down2_3 = Downsample(int(dim * 2 ** 1))
        self.encoder_level3 = blocks(int(dim * 2 ** 2), heads[2], num_blocks[2])
        self.down3_4 = Downsample(int(dim * 2 ** 2))
        self.latent = blocks(int(dim * 2 ** 3), heads[3], num_blocks[3])
        self.up4_3 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level3 = nn.Conv2d(int(dim * 2 ** 1) + 192, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.noise_level3 = TransformerBlock(int(dim * 2 ** 2) + 512, heads[2], ffn_expansion_factor, bias, LayerNorm_type)
        self.reduce_noise_level3 = nn.Conv2d(int(dim * 2 ** 2) + 512, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level3 = blocks(int(dim * 2 ** 2), heads[2], num_blocks[2])
        self.up3_2 = Upsample(int(dim * 2 ** 2))
        self.reduce_chan_level2 = nn.Conv2d(int(dim * 2 ** 2), int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.noise_level2 = TransformerBlock(int(dim * 2 ** 1) + 224, heads[2], ffn_expansion_factor, bias, LayerNorm_type)
        self.reduce_noise_level2 = nn.Conv2d(int(dim * 2 ** 1) + 224, int(dim * 2 ** 2), kernel_size=1, bias=bias)
        self.decoder_level2 = blocks(int(dim * 2 ** 1), heads[1], num_blocks[1])
        self.up2_1 = Upsample(int(dim * 2 ** 1))
        self.noise_level1 = TransformerBlock(int(dim * 2 ** 1) + 64, heads[2], ffn_expansion_factor, bias, LayerNorm_type)
        self.reduce_noise_level1 = nn.Conv2d(int(dim * 2 ** 1) + 64, int(dim * 2 ** 1), kernel_size=1, bias=bias)
        self.decoder_level1 = blocks(int(dim * 2 ** 1), heads[0], num_blocks[0])
        self.refinement = blocks(int(dim * 2 ** 1), heads[0], num_refinement_blocks)
        self.output = nn.Conv2d(int(dim * 2 ** 1), out_channels, kernel_size=3, stride=1, padding=1, bias=bias)

    def _prompted(self, x, prompt, noise, reduce):
        x = DF.concat_channels(x, prompt(x))
        return DF.conv_nobias(noise(x), reduce.weight)

    def forward(self, inp_img, hook: bool = False):
        e1 = self.encoder_level1(self.patch_embed(inp_img))
        e2 = self.encoder_level2(self.down1_2(e1))
        e3 = self.encoder_level3(self.down2_3(e2))
        latent = self.latent(self.down3_4(e3))
        if self.decoder:
            latent = self._prompted(latent, self.prompt3, self.noise_level3, self.reduce_noise_level3)
        d3 = DF.conv_nobias(DF.concat_channels(self.up4_3(latent), e3), self.reduce_chan_level3.weight)
        d3 = self.decoder_level3(d3)
        if self.decoder:
            d3 = self._prompted(d3, self.prompt2, self.noise_level2, self.reduce_noise_level2)
        d2 = DF.conv_nobias(DF.concat_channels(self.up3_2(d3), e2), self.reduce_chan_level2.weight)
        d2 = self.decoder_level2(d2)
        if self.decoder:
            d2 = self._prompted(d2, self.prompt1, self.noise_level1, self.reduce_noise_level1)
        if hook:
            return None   # the reference falls off the end of forward() here (:497-518)
        d1 = self.decoder_level1(DF.concat_channels(self.up2_1(d2), e1))
        return DF.conv3x3_out(self.refinement(d1), self.output.weight, None, inp_img)
