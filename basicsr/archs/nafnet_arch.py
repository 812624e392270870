"""NAFNet for the DCPT restoration encoder, MI355X-native.

Same registry names, constructor kwargs, ``forward(inp, hook=False)`` contract, module tree and
``state_dict`` keys/shapes as the reference (basicsr/archs/nafnet_arch.py), so checkpoints load
with ``strict=True`` and forward hooks on ``decoder{i}`` see the block-group outputs.  The math does
not run through torch ops: every NAFBlock is ONE autograd node backed by ``dcpt_nafblock_fwd/bwd``
(LayerNorm as its own bandwidth kernel whose output feeds the MFMA GEMMs as a plain LDS-DMA operand --
or, at C <= 128, inside the producing GEMM's epilogue --, depthwise 3x3 + SimpleGate + pooling in one
pass, SCA / residual scales / SimpleGate in GEMM epilogues; LABNOTES.md section 4), and the convs between
blocks are ``dcpt_conv3x3_*``, ``dcpt_down2x2_*`` and ``dcpt_up_ps_*``.  Feature maps are channels_last
(NHWC) tensors.

``act_dtype="bf16"`` (an extension; the reference computes in fp32 only) keeps every feature map in bfloat16
storage from the intro conv's output to the ending conv's input -- NAFBlocks (``dcpt_nafblock_fwd_bf16/bwd_bf16``)
and the layers between them (``dcpt_conv3x3_in/out_*_bf16``, ``dcpt_down2x2_*_bf16``, ``dcpt_up_ps_*_bf16``) -- with
fp32 accumulation; images, all parameters and their gradients stay fp32 and there are no cast kernels.  Needs
``width % 8 == 0``; forward hooks on the block groups see bf16 feature maps.

The child ``nn.Conv2d`` / ``LayerNorm2d`` modules below only OWN the parameters (for state-dict and
optimizer compatibility); they are never called on the fused path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from basicsr.utils.registry import ARCH_REGISTRY
from dcpt_amd import functional as DF

from .arch_util import AvgPool2d, Local_Base


class LayerNorm2d(nn.Module):
    """Per-pixel channel LayerNorm (reference nafnet_arch.py:56-64); weight/bias shape (C,)."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.eps = eps

    def forward(self, x):
        return DF.layernorm2d(x, self.weight, self.bias, self.eps)


class SimpleGate(nn.Module):
    """x[:, :c] * x[:, c:] (reference nafnet_arch.py:77-80).  Inside NAFBlock it is fused into the
    depthwise-conv kernel and the conv5 GEMM loader; the module exists for API parity."""

    def forward(self, x):
        x1, x2 = x.chunk(2, dim=1)
        return x1 * x2


class NAFBlock(nn.Module):
    """reference nafnet_arch.py:83-186.  drop_out_rate must be 0 (the reference never sets it)."""

    def __init__(self, c, DW_Expand=2, FFN_Expand=2, drop_out_rate=0.0):
        super().__init__()
        if DW_Expand != 2 or FFN_Expand != 2:
            raise NotImplementedError("dcpt_amd NAFBlock kernels implement DW_Expand = FFN_Expand = 2")
        if drop_out_rate > 0.0:
            raise NotImplementedError("dcpt_amd NAFBlock kernels implement drop_out_rate = 0 (the reference default)")
        if c % 4:
            raise ValueError(f"NAFBlock width {c} must be a multiple of 4 for the NHWC float4 kernels")
        dw = 2 * c
        self.conv1 = nn.Conv2d(c, dw, 1, bias=True)
        self.conv2 = nn.Conv2d(dw, dw, 3, padding=1, groups=dw, bias=True)
        self.conv3 = nn.Conv2d(dw // 2, c, 1, bias=True)
        self.sca = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(dw // 2, dw // 2, 1, bias=True))
        self.sg = SimpleGate()
        self.conv4 = nn.Conv2d(c, 2 * c, 1, bias=True)
        self.conv5 = nn.Conv2d(c, c, 1, bias=True)
        self.norm1 = LayerNorm2d(c)
        self.norm2 = LayerNorm2d(c)
        self.dropout1 = nn.Identity()
        self.dropout2 = nn.Identity()
        self.beta = nn.Parameter(torch.zeros((1, c, 1, 1)), requires_grad=True)
        self.gamma = nn.Parameter(torch.zeros((1, c, 1, 1)), requires_grad=True)

    def fused_params(self):
        return {
            "norm1_w": self.norm1.weight, "norm1_b": self.norm1.bias,
            "conv1_w": self.conv1.weight, "conv1_b": self.conv1.bias,
            "conv2_w": self.conv2.weight, "conv2_b": self.conv2.bias,
            "conv3_w": self.conv3.weight, "conv3_b": self.conv3.bias,
            "sca_w": self.sca[1].weight, "sca_b": self.sca[1].bias,
            "norm2_w": self.norm2.weight, "norm2_b": self.norm2.bias,
            "conv4_w": self.conv4.weight, "conv4_b": self.conv4.bias,
            "conv5_w": self.conv5.weight, "conv5_b": self.conv5.bias,
            "beta": self.beta, "gamma": self.gamma,
        }

    # bf16-storage mode (NAFNetBaseline.set_act_dtype): the block runs on the bf16 kernels and hands bf16 on -- the layers between
    # the groups and the forward hooks on ``decoder{i}`` (the DCPT taps) see bf16 feature maps (module docstring)
    act_bf16 = False

    def local_sca(self):
        pool = self.sca[0]
        return isinstance(pool, AvgPool2d) and pool.kernel_size is not None

    def forward(self, inp):
        pool = self.sca[0]
        if isinstance(pool, AvgPool2d) and pool.kernel_size is not None:
            k1, k2 = int(pool.kernel_size[0]), int(pool.kernel_size[1])
            if not (k1 >= inp.shape[-2] and k2 >= inp.shape[-1]):  # arch_util.py:352-353: window covers the map -> global mean
                if self.act_bf16:
                    raise NotImplementedError("the TLSC (local-mean SCA) block is fp32 only: build NAFNet with act_dtype='fp32'")
                return DF.nafblock_local(inp, self.fused_params(), k1, k2)
        if self.act_bf16:   # bf16 in / out (an fp32 input -- a block used on its own -- is cast once)
            if getattr(self, "_packed_bf16", None) is None:
                self._packed_bf16 = DF.PackedWeightsBf16()   # operand copies of the weights, refreshed when the parameters change
            return DF.nafblock_bf16(DF.to_bf16(inp), self.fused_params(), self._packed_bf16)
        return DF.nafblock(inp, self.fused_params())


class _BlockGroup(nn.Sequential):
    """A level's NAFBlocks (an ``nn.Sequential`` in the reference: same state-dict keys, same forward-hook targets)."""


class _Down(nn.Conv2d):
    """Conv2d(c, 2c, 2, 2) run as a gathered MFMA GEMM (reference nafnet_arch.py:230)."""

    def forward(self, x):
        return DF.down2x2(x, self.weight, self.bias)


class _UpConv(nn.Conv2d):
    """The 1x1 conv of an ``ups[i]`` Sequential; the PixelShuffle and the skip add are fused into its
    epilogue by NAFNetBaseline.forward (reference nafnet_arch.py:238-242, :264-265)."""


_ACT_DTYPES = ("fp32", "bf16", "bf16_tail32", "bf16_edge32")


@ARCH_REGISTRY.register()
class NAFNetBaseline(nn.Module):
    def __init__(self, img_channel=3, width=16, middle_blk_num=1, enc_blk_nums=[], dec_blk_nums=[], window_size=8, act_dtype="fp32",
                 gemm_precision=None):
        super().__init__()
        if act_dtype not in _ACT_DTYPES:
            raise ValueError(f"act_dtype must be one of {_ACT_DTYPES}, got {act_dtype!r}")
        if gemm_precision not in (None, "fp32", "bf16x3"):
            raise ValueError(f"gemm_precision must be 'fp32' or 'bf16x3', got {gemm_precision!r}")
        # ``network_g.gemm_precision`` (this repo's extension, fp32 storage only): "fp32" = exact fp32 MFMA, the reference's arithmetic;
        # "bf16x3" = the wide 1 x 1 convs as split-operand products on the bf16 matrix pipe (fp32-class, LABNOTES.md 4d).  Absent: the
        # process default (dcpt_amd.functional.set_gemm_precision), which is "fp32" unless a caller changed it.
        self.gemm_precision = gemm_precision
        self.intro = nn.Conv2d(img_channel, width, 3, padding=1, bias=True)
        self.ending = nn.Conv2d(width, img_channel, 3, padding=1, bias=True)
        self.encoders = nn.ModuleList()
        self.middle_blks = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.downs = nn.ModuleList()

        chan = width
        for num in enc_blk_nums:
            self.encoders.append(_BlockGroup(*[NAFBlock(chan) for _ in range(num)]))
            self.downs.append(_Down(chan, 2 * chan, 2, 2))
            chan *= 2
        self.middle_blks = _BlockGroup(*[NAFBlock(chan) for _ in range(middle_blk_num)])
        for i, num in enumerate(dec_blk_nums):
            self.ups.append(nn.Sequential(_UpConv(chan, chan * 2, 1, bias=False), nn.PixelShuffle(2)))
            chan //= 2
            setattr(self, f"decoder{i}", _BlockGroup(*[NAFBlock(chan) for _ in range(num)]))
        self._n_dec = len(dec_blk_nums)
        self._width = width
        self.set_act_dtype(act_dtype)

    def set_act_dtype(self, act_dtype):
        """'fp32' (the reference's arithmetic), 'bf16' (bf16 storage of every feature map, fp32 accumulate), 'bf16_tail32' (bf16 storage
        up to the last decoder group; that group -- full resolution, ``width`` channels -- and the ending conv run in fp32: the SET mean
        of the PSNR stays inside the 0.01-dB gate, single images scatter up to 0.016 dB -- outside it) or 'bf16_edge32' (additionally the intro conv and
        the FIRST encoder group in fp32, i.e. everything at full resolution: the skip connection into the last decoder group is then an
        fp32 tensor end to end -- every single image inside the gate; tests/test_gpu_configs.py::test_psnr_bf16_storage_vs_fp32)"""
        if act_dtype not in _ACT_DTYPES:
            raise ValueError(f"act_dtype must be one of {_ACT_DTYPES}, got {act_dtype!r}")
        if act_dtype != "fp32" and self._width % 8:
            # 16-byte bf16 rows at every level (the up layers' 2c -> c/2 PixelShuffle cells included: c = width * 2^k)
            raise ValueError(f"act_dtype='bf16' needs width % 8 == 0 (16-byte bf16 channel vectors), got width={self._width}")
        self.act_dtype = act_dtype
        self.__dict__.pop("_bf16_blocks", None)
        for m in self.modules():
            if isinstance(m, NAFBlock):
                m.act_bf16 = act_dtype != "fp32"
        if act_dtype in ("bf16_tail32", "bf16_edge32") and self._n_dec > 0:
            for m in getattr(self, f"decoder{self._n_dec - 1}"):
                m.act_bf16 = False
        if act_dtype == "bf16_edge32" and len(self.encoders) > 0:
            for m in self.encoders[0]:
                m.act_bf16 = False

    def forward(self, inp, hook=False):
        with DF.gemm_precision(self.gemm_precision):   # (the backward of every node built here runs under the same mode)
            return self._forward(inp, hook)

    def _forward(self, inp, hook=False):
        # bf16 storage: the intro conv emits bf16 features and every layer up to the ending conv's input stays bf16 (forward hooks on
        # the block groups then see bf16 feature maps; the classifier head takes either dtype)
        if self.act_dtype != "fp32" and inp.is_cuda:
            # the operand copies of every bf16 block's weights in one go where they are stale (after an optimizer step: 5 launches
            # instead of 36); the blocks then find their pack current
            blocks = self.__dict__.get("_bf16_blocks")
            if blocks is None:   # (a plain attribute, not a registered submodule list; rebuilt by set_act_dtype)
                blocks = [m for m in self.modules() if isinstance(m, NAFBlock) and m.act_bf16 and not m.local_sca()]
                for m in blocks:
                    if getattr(m, "_packed_bf16", None) is None:
                        m._packed_bf16 = DF.PackedWeightsBf16()
                self.__dict__["_bf16_blocks"] = blocks
            DF.pack_blocks_bf16([(m._packed_bf16, m.fused_params()) for m in blocks])
        edge32 = self.act_dtype == "bf16_edge32" and len(self.encoders) > 0 and inp.is_cuda
        x = DF.conv3x3_in(inp, self.intro.weight, self.intro.bias, out_bf16=self.act_dtype != "fp32" and not edge32)
        encs = []
        for lvl, (encoder, down) in enumerate(zip(self.encoders, self.downs)):
            x = encoder(x)
            if edge32 and lvl == 0:
                encs.append(x)
                x = down(DF.to_bf16(x))   # (the fp32 first group hands bf16 to the levels below; its own output stays fp32 for the skip)
            elif x.is_cuda and torch.is_grad_enabled() and x.requires_grad:
                # one autograd node for the group output's two consumers (down layer, skip): their gradients are summed in the down layer's
                # data-gradient GEMM instead of by an `add` pass of autograd's (DF._DownSkipFn)
                x, skip = DF.down2x2_skip(x, down.weight, down.bias)
                encs.append(skip)
            else:
                encs.append(x)
                x = down(x)
        x = self.middle_blks(x)
        for i, (up, enc_skip) in enumerate(zip(self.ups, encs[::-1])):
            if edge32 and i == self._n_dec - 1:
                x = DF.to_f32(x)   # fp32 from here on: the up layer adds the fp32 skip of the first encoder group
            x = DF.up_ps(x, up[0].weight, enc_skip)  # conv1x1 + PixelShuffle(2) + skip add, one kernel
            if self.act_dtype == "bf16_tail32" and i == self._n_dec - 1:
                x = DF.to_f32(x)   # the last group and the ending conv in fp32 (one cast of a width-channel map)
            x = getattr(self, f"decoder{i}")(x)
        if not hook:
            return DF.conv3x3_out(x, self.ending.weight, self.ending.bias, inp)
        return None


@ARCH_REGISTRY.register()
class NAFNet(Local_Base, NAFNetBaseline):
    """TLSC test-time variant (reference nafnet_arch.py:277-288): every SCA global mean becomes a local box mean of
    1.5 x the training patch size (scaled per level); inference only."""

    def __init__(self, *args, train_size=(1, 3, 128, 128), fast_imp=False, **kwargs):
        Local_Base.__init__(self)
        NAFNetBaseline.__init__(self, *args, **kwargs)
        N, C, H, W = train_size
        base_size = (int(H * 1.5), int(W * 1.5))
        self.eval()
        with torch.no_grad():
            self.convert(base_size=base_size, train_size=train_size, fast_imp=fast_imp)

    def _assign_local_kernels(self, train_size):
        h, w = train_size[-2], train_size[-1]
        n_enc = len(self.encoders)
        groups = [(enc, i) for i, enc in enumerate(self.encoders)] + [(self.middle_blks, n_enc)]
        groups += [(getattr(self, f"decoder{i}"), n_enc - 1 - i) for i in range(self._n_dec)]
        for seq, level in groups:
            for blk in seq:
                blk.sca[0].set_kernel_from_feature(h >> level, w >> level)
