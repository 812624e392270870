"""Degradation-classifier head of DCPT, MI355X-native.

Same registry names, constructor kwargs, ``forward(lq, features)`` contract and ``state_dict`` keys/shapes as
the reference's ``PromptIR_NoImg_DC`` (basicsr/archs/degrad_classify_arch.py:558-641; 97 keys for
feature_dims of length 4 with 2 blocks) and ``PromptIR_DC`` (:480-555: the same head seeded by a 7x7 stride-2
embedding of ``lq``, so its feature maps start at half the image resolution).  Every conv -> LayerNorm(channels) -> (+shortcut) -> ReLU group is one
autograd node backed by ``dcpt_conv_ln_fwd/bwd`` (1x1 and dense 3x3 convolutions run as fp32 MFMA GEMMs, the 3x3
as an implicit GEMM), the downsample layers by ``dcpt_conv1x1_pool_relu_*``, the softmax feature mixing by
``dcpt_mix_*`` and the head by ``dcpt_meanpool_fc_*``.  Child modules only own the parameters.

``act_dtype="bf16"`` (an extension; the reference computes in fp32 only): the bottleneck groups and the downsample layers run on
the bf16-storage kernels (``dcpt_conv_ln_*_bf16``, ``dcpt_conv1x1_pool_relu_*_bf16``: bf16 activations, fp32 parameters and
accumulation); the mixing step and mean + Linear stay fp32 behind casts.  Needs feature_dims that are multiples of 8.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from basicsr.utils.registry import ARCH_REGISTRY
from dcpt_amd import functional as DF


class LayerNorm(nn.Module):
    """channels_first LayerNorm parameters (reference :17-44, eps 1e-6); applied inside the fused conv op."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_first"):
        super().__init__()
        if data_format != "channels_first":
            raise NotImplementedError("only channels_first is on the DCPT path")
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.normalized_shape = (normalized_shape,)


class Conv2d(nn.Conv2d):
    """conv + norm (+ activation) holder (reference :69-103); ``fused`` runs conv -> LN -> [+res] -> [ReLU]."""

    def __init__(self, *args, norm=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm
        self._pack = DF.PackedConvBf16()   # the weight's bf16 operand images, cached across the calls of an optimizer step (bf16 path)

    def fused(self, x, res=None, relu=True):
        if x.dtype == torch.bfloat16:
            return DF.conv_ln_bf16(x, self.weight, self.norm.weight, self.norm.bias, res, relu, packed=self._pack if x.is_cuda else None)
        return DF.conv_ln(x, self.weight, self.norm.weight, self.norm.bias, res, relu)


class BottleneckBlock(nn.Module):
    """reference :132-243 restricted to what the DCPT head instantiates: in == out channels (identity shortcut),
    stride 1, norm "LN", bias-free convs."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, norm="LN"):
        super().__init__()
        if in_channels != out_channels or norm != "LN":
            raise NotImplementedError("the DCPT head uses identity-shortcut LN bottlenecks only")
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, 1
        self.shortcut = None
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, bias=False, norm=LayerNorm(bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, padding=1, bias=False,
                            norm=LayerNorm(bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False, norm=LayerNorm(out_channels))
        for layer in (self.conv1, self.conv2, self.conv3):  # c2_msra_fill (fvcore): kaiming_normal fan_out / relu
            nn.init.kaiming_normal_(layer.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        if x.is_cuda:   # one autograd node: the shortcut's gradient is summed inside conv1's data-gradient GEMM (DF._BottleneckFn)
            c1, c2, c3 = self.conv1, self.conv2, self.conv3
            return DF.bottleneck(x, c1.weight, c1.norm.weight, c1.norm.bias, c2.weight, c2.norm.weight, c2.norm.bias,
                                 c3.weight, c3.norm.weight, c3.norm.bias, packs=(c1._pack, c2._pack, c3._pack))
        out = self.conv1.fused(x, relu=True)
        out = self.conv2.fused(out, relu=True)
        return self.conv3.fused(out, res=x, relu=True)  # relu(LN(conv3) + shortcut)


class _Downsample(nn.Sequential):
    """Conv2d(1x1, bias=False) -> MaxPool2d(2,2) -> ReLU as one op (reference :596-602)."""

    def forward(self, x):
        if x.dtype == torch.bfloat16:
            return DF.conv1x1_pool_relu_bf16(x, self[0].weight, packed=self._conv_pack() if x.is_cuda else None)
        return DF.conv1x1_pool_relu(x, self[0].weight)

    def _conv_pack(self):
        pk = self.__dict__.get("_pack")
        if pk is None:
            pk = self.__dict__["_pack"] = DF.PackedConvBf16()
        return pk


class _DCHead(nn.Module):
    """stages shared by the two heads (reference :497-552 and :576-619 build the same modules)"""

    def _build_stages(self, feature_dims, num_res_blocks, num_classes):
        self.feature_dims = list(feature_dims)
        self.bottleneck_layers = nn.ModuleList()
        self.downsample_layers = nn.ModuleList()
        for l, dim in enumerate(self.feature_dims):
            self.bottleneck_layers.append(nn.Sequential(*[
                BottleneckBlock(dim, dim, bottleneck_channels=int(dim * 2), norm="LN") for _ in range(num_res_blocks)]))
            nxt = self.feature_dims[l + 1] if l < len(self.feature_dims) - 1 else dim
            self.downsample_layers.append(_Downsample(nn.Conv2d(dim, nxt, 1, bias=False), nn.MaxPool2d(2, 2), nn.ReLU()))
        last = self.feature_dims[-1]
        self.last_stage = nn.Sequential(*[
            BottleneckBlock(last, last, bottleneck_channels=int(last * 2), norm="LN") for _ in range(num_res_blocks)])
        self.mixing_weights = nn.Parameter(torch.ones(len(self.bottleneck_layers)), requires_grad=True)
        self.fc = nn.Linear(last, num_classes)

    act_dtype = "fp32"

    def _set_act_dtype(self, act_dtype):
        if act_dtype not in ("fp32", "bf16"):
            raise ValueError(f"act_dtype must be 'fp32' or 'bf16', got {act_dtype!r}")
        if act_dtype == "bf16" and any(d % 8 for d in self.feature_dims):
            raise ValueError("act_dtype='bf16' needs feature_dims that are multiples of 8")
        self.act_dtype = act_dtype

    def _packed_convs(self):
        """[(cache, weight)] of every conv of the head: refreshed together (three launches) in front of a bf16 forward"""
        out = []
        for m in self.modules():
            if isinstance(m, Conv2d):
                out.append((m._pack, m.weight))
            elif isinstance(m, _Downsample):
                out.append((m._conv_pack(), m[0].weight))
        return out

    def _run_stages(self, x, features):
        bf = self.act_dtype == "bf16"
        if bf and len(features) and features[0].is_cuda:
            DF.pack_convs_bf16(self._packed_convs())
        for i, feature in enumerate(features):
            if bf and feature.is_cuda and feature.dtype == torch.bfloat16 and (x is None or x.dtype == torch.bfloat16):
                # taps and stage outputs both bf16 (the all-bf16 DCPT step): mixed in fp32, stored once -- no cast passes
                x = DF.mix(x, feature, self.mixing_weights, i)
            else:
                x = DF.mix(None if x is None else DF.to_f32(x), DF.to_f32(feature), self.mixing_weights, i)   # fp32: one pass per stage
                if bf:
                    x = DF.to_bf16(x)
            x = self.bottleneck_layers[i](x)
            x = self.downsample_layers[i](x)
        x = self.last_stage(x)
        if bf and x.is_cuda and x.dtype == torch.bfloat16:
            return DF.meanpool_fc(x, self.fc.weight, self.fc.bias)   # pooled in fp32 straight from the bf16 map
        return DF.meanpool_fc(DF.to_f32(x), self.fc.weight, self.fc.bias)


@ARCH_REGISTRY.register()
class PromptIR_NoImg_DC(_DCHead):
    def __init__(self, feature_dims, num_res_blocks=2, num_classes=3, downsample=False, act_dtype="fp32"):
        super().__init__()
        if downsample:
            raise NotImplementedError("downsample=True (token inputs) is not on the DCPT path")
        self.downsample = downsample
        self._build_stages(feature_dims, num_res_blocks, num_classes)
        self._set_act_dtype(act_dtype)

    def forward(self, lq, features):
        """``lq`` is accepted and ignored, exactly like the reference (:621)."""
        return self._run_stages(None, features)


@ARCH_REGISTRY.register()
class PromptIR_DC(_DCHead):
    """reference :480-555.  ``features[i]`` must have the resolution of the embedded image at stage i (H/2, H/4, ...);
    like the reference, anything else fails (there with a broadcast error in ``lq_feats + w * feature``)."""

    def __init__(self, feature_dims, num_res_blocks=2, num_classes=3):
        super().__init__()
        self.conv_embed = nn.Sequential(nn.Conv2d(3, feature_dims[0], 7, 2, 3), LayerNorm(feature_dims[0]))
        self._build_stages(feature_dims, num_res_blocks, num_classes)

    def forward(self, lq, features):
        conv, norm = self.conv_embed[0], self.conv_embed[1]
        x = DF.conv_embed_ln(lq, conv.weight, conv.bias, norm.weight, norm.bias, stride=2, pad=3)
        for i, feature in enumerate(features):
            want = (x.shape[0], self.feature_dims[i], x.shape[2] >> i, x.shape[3] >> i)
            if tuple(feature.shape) != want:
                raise RuntimeError(f"PromptIR_DC: feature {i} has shape {tuple(feature.shape)}, the embedded image needs {want}")
        return self._run_stages(x, features)
